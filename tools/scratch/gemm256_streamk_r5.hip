// Large-tile variant of the dense projection for prefill-sized problems (N % 256 == 0, K % 64 == 0; p3v_gemm decides
// which problems it gets):
//   256(M) x 256(N) x 64(K) tile, 512 threads = 8 waves as 2(M) x 4(N), each wave 128 x 64 =
//   8 x 4 v_mfma_f32_16x16x32_bf16 accumulators (128 registers), 2.67 MFMAs per LDS fragment read
//   (the 128x128 kernel: 2.0) and 25 % less global->LDS traffic per flop.
//   K-tile t lives in LDS buffer t&1 (2 x 64 KiB); a tile is four PHASES (one 64x32 accumulator quadrant = 16
//   MFMAs each).  The four 16-KiB half-tiles of tile t+1 are requested by LDS-DMA in the first two phases of
//   tile t, each batch AFTER that phase's fragment reads are issued: measured against one half-tile per phase
//   (the last one then has only 16 MFMAs to land before the vmcnt(0)) +12 % at 4096^3 (1170 -> 1310 TF/s), against
//   all four up front +7 % (eight DMA issues ahead of the first fragment reads delay the first MFMA).  The fragment
//   reads are software-pipelined: the fragments of phase p+1 are requested before the MFMAs of phase p (second
//   A-fragment buffer, 256 registers, no spill), +1.5-2 % at 4096^3 / 8192^3.  One vmcnt(0) + barrier per K-tile.
//   No barrier inside a tile: the waves de-phase, one wave's fragment reads overlap another's MFMAs.
//   Same XOR-swizzled LDS image and the same epilogues as p3v_gemm.hip (through a wave-private LDS tile).
//
// Round 5: STREAM-K.  The launch is one persistent workgroup per CU.  Output tiles come in a fixed order (XCD-aware, below);
// the first `sk_tiles` of them are not handed out whole but as a run of K-ITERATIONS: workgroup position q takes iterations
// [q * I / G, (q + 1) * I / G) of the I = sk_tiles * (K / 64) the region holds, so every CU gets the same work to within
// one K-tile whatever the tile count is (2531 x 9216 is 360 tiles = 1.41 rounds of 256 CUs; before: whole rounds of big
// tiles + a second launch of 128 x 128 tiles for the remaining rows, and N = 3072 ran on the small tiles altogether).
// The remaining tiles (a multiple of the grid) run whole, one per workgroup and round, after the shared region.
// A tile cut by a range boundary is finished by whichever of its contributors ARRIVES LAST (a ticket per tile): the others
// write their 256 x 256 fp32 accumulators to a slab of the caller's workspace (write-through stores, accumulator layout,
// fully coalesced) and move on; the last one adds the slabs IN POSITION ORDER (its own registers take their place in
// that order, so the sum does not depend on who was last: same bits every launch) and runs the epilogue.  Nobody waits
// for a workgroup that has not started: the only wait is the last arriver's for contributors that already drew a ticket
// (they are resident and storing), so the scheme needs no co-residency and no dispatch order.  Flags are reset by the
// reducer (zero between launches: graph-replayable); the workspace header must be zeroed once by its owner.
#include <stdlib.h>

#include "p3v_common.h"

#define TM 256
#define TN 256
#define TK 64
#define HALF_BYTES (128 * TK * 2)     // 16 KiB
#define BUF_BYTES (4 * HALF_BYTES)    // A0 A1 B0 B1
#define GEMM256_LDS (2 * BUF_BYTES)   // 128 KiB (epilogue: 8 waves x 64x68 fp32 = 136 KiB would not fit -> four passes of 32 rows)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Gemm256P {
  const bf16_t* A; const bf16_t* W; void* out; const bf16_t* bias; const void* resid;
  int M, N, K, lda, ldw, ldo;
  int sk_tiles;            // tiles [0, sk_tiles) of the tile order are shared out by K-iterations; the rest run whole
  int* sk_flags;           // arrive[P3V_SK_MAX_TILES], done[P3V_SK_MAX_TILES]: zero between launches
  float* sk_slabs;         // [2 * gridDim.x] partial tiles of 256 x 256 fp32 (slot 2q / 2q+1: position q's first / last item)
};

struct Tile256 { int m0, n0; int a_off[2][2], b_off[2][2]; __amdgpu_buffer_rsrc_t rs_a; };
struct Item256 { int tile, k0, k1; };   // K-tiles [k0, k1) of output tile `tile` (< 0: nothing left)

__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

#ifdef P3V_G256_DEBUG                                            // tools/gemm256_timeline.py: per-wave stamps inside the K loop of ONE workgroup
__device__ unsigned long long p3v_g256dbg[8 * 64 * 8];
#define G_S(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(g_s[k])); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G_FLUSH(it) do { if (blockIdx.x == P3V_G256_DEBUG && (threadIdx.x & 63) == 0 && (it) < 64) { for (int k_ = 0; k_ < 8; ++k_) p3v_g256dbg[((threadIdx.x >> 6) * 64 + (it)) * 8 + k_] = g_s[k_]; } } while (0)
extern "C" int p3v_g256dbg_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_g256dbg), sizeof(unsigned long long) * 8 * 64 * 8) == hipSuccess ? 0 : -1;
}
#else
#define G_S(k) do { } while (0)
#define G_FLUSH(it) do { } while (0)
#endif

template <int EPI>
__global__ void __launch_bounds__(512, 1) k_gemm256(Gemm256P p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: LDS-DMA bases stay in SGPRs
  const int wr = wave >> 2, wc = wave & 3;
  constexpr int n_out_tile = SILU ? TN / 2 : TN;
  const int gx = (p.N + n_out_tile - 1) / n_out_tile, gy = (p.M + TM - 1) / TM, nwg = gx * gy;
  const int nk = p.K / TK;

  // ---- work of this workgroup.  The dispatcher places workgroup b on XCD b % 8 (speed only, never correctness): position
  // q = (b % 8) * (G / 8) + b / 8 gives each XCD a contiguous run of positions, hence of tiles -- the tiles resident on one XCD
  // share A row panels and W panels through that XCD's 4 MiB L2.  (G is a multiple of 8.)
  const int G = gridDim.x;
  const int pos = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const int I = p.sk_tiles * nk;                                  // K-iterations of the shared region
  auto sk_start = [&](int q) { return (int)((unsigned)q * (unsigned)I / (unsigned)G); };   // (G + 1) * I < 2^31: launcher
  int it = sk_start(pos);
  const int it_begin = it, it_end = sk_start(pos + 1);
  int dp = p.sk_tiles + pos;                                      // whole tiles: dp, dp + G, ...
  auto next_item = [&](Item256& o) {
    if (it < it_end) {
      o.tile = it / nk;
      o.k0 = it - o.tile * nk;
      o.k1 = min(nk, o.k0 + it_end - it);
      it += o.k1 - o.k0;
    } else if (dp < nwg) {
      o.tile = dp; o.k0 = 0; o.k1 = nk;
      dp += G;
    } else {
      o.tile = -1; o.k0 = o.k1 = 0;
    }
    // (wave-uniform by construction; said explicitly, or hipcc carries the tile's buffer descriptor in vector registers and
    //  wraps every LDS-DMA request in a waterfall loop)
    o.tile = __builtin_amdgcn_readfirstlane(o.tile), o.k0 = __builtin_amdgcn_readfirstlane(o.k0), o.k1 = __builtin_amdgcn_readfirstlane(o.k1);
  };

  // The K-tile stream runs ACROSS item seams: during the last K-tile of an item the first K-tile of the workgroup's NEXT item is
  // requested into the other LDS buffer, and the epilogue (staged through the buffer that was just consumed) runs under that DMA
  // -- the 64-KiB cold fetch that opened every tile (HBM latency + 4 x 16 KiB, ~2 us) was 15-20 % of a K = 1024 tile (the ViT
  // projections) and 5 % at K = 3072.
  using Tile = Tile256;   // (declared outside the template: a builtin called with a member of a DEPENDENT type is only checked at
                          //  instantiation, fails there in the HOST pass -- no such builtin -- and hipcc silently drops the host stub)
  const int srow = tid >> 3, schunk = tid & 7;
  auto make_tile = [&](int id, Tile& t) {
    constexpr int BAND = 4;
    const int band = id / (BAND * gx), in_band = id % (BAND * gx);
    const int rows = min(BAND, gy - band * BAND);
    const int m_t = band * BAND + in_band % rows, n_t = in_band / rows;
    t.m0 = __builtin_amdgcn_readfirstlane(m_t * TM), t.n0 = __builtin_amdgcn_readfirstlane(n_t * n_out_tile);
    // DMA sources: half-tile h (128 rows), instruction q (64 rows), this thread: row tid/8, chunk tid%8.
    // Buffer addressing (SGPR descriptor + 32-bit per-lane byte offset + SGPR K offset): a request is `s_mov m0` +
    // `buffer_load_dwordx4 ... offen lds` with no vector ALU work at all.
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int rr = h * 128 + qq * 64 + srow;
        const int sw = (schunk ^ (rr & 7)) * 8;
        const int ar = min(t.m0 + rr, p.M - 1) - t.m0;
        t.a_off[h][qq] = (ar * p.lda + sw) * 2;
        int br;
        if (SILU) {                                   // wave column group wcol (64 tile rows) = 32 gate + 32 up rows
          const int wcol = rr >> 6, ni = (rr & 63) >> 4, c = rr & 15;
          br = min(t.n0 + wcol * 32 + (ni & 1) * 16 + c, p.N - 1) + (ni >> 1) * p.N;
        } else {
          br = min(t.n0 + rr, p.N - 1);
        }
        t.b_off[h][qq] = (int)(((unsigned)br * (unsigned)p.ldw + (unsigned)sw) * 2u);   // < 2^32: checked by the launcher
      }
    t.rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)t.m0 * p.lda), 0, 0xffffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
  auto dma_half = [&](const Tile& t, int which, int kt, int buf) {   // which: 0 A0, 1 A1, 2 B0, 3 B1
    unsigned char* base = smem + buf * BUF_BYTES + which * HALF_BYTES + wave * 1024;
    const int h = which & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const __amdgpu_buffer_rsrc_t rs = which < 2 ? t.rs_a : rs_w;
      const int vo = which < 2 ? t.a_off[h][q] : t.b_off[h][q];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(base + q * 8192), 16, vo, kt * (TK * 2), 0, 0);
    }
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  Item256 cur, nxt;
  next_item(cur);
  if (cur.tile < 0) return;
  Tile cur_t;
  make_tile(cur.tile, cur_t);
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) dma_half(cur_t, w4, cur.k0, 0);
  int gk = 0;                                                     // K-tiles consumed so far: LDS buffer = gk & 1
  for (;;) {
  next_item(nxt);
  const int m0 = cur_t.m0, n0 = cur_t.n0;
  const bool has_next = nxt.tile >= 0;
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#ifdef P3V_G256_DEBUG
  unsigned long long g_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int kt = cur.k0; kt < cur.k1; ++kt, ++gk) {
    G_S(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G_S(1);
    __syncthreads();
    G_S(2);
    const bool last = kt + 1 == cur.k1;
    const bool more = !last || has_next;                          // something to prefetch: this item's next K-tile, or the next item's first
    const int nb = (gk + 1) & 1;
    const unsigned char* ta = smem + (gk & 1) * BUF_BYTES + wr * HALF_BYTES;                      // this wave's A half
    const unsigned char* tb = smem + (gk & 1) * BUF_BYTES + (2 + (wc >> 1)) * HALF_BYTES + (wc & 1) * 64 * 128;  // its 64 B rows
    bf16x8_t af[4][2], af1[4][2], bf0[2][2], bf1[2][2];
    auto read_a_to = [&](int sub, bf16x8_t (&dst)[4][2]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int r = sub * 64 + i * 16 + frow;
          dst[i][kk] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
        }
    };
    auto read_b = [&](int sub, bf16x8_t (&bf)[2][2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int r = sub * 32 + j * 16 + frow;
          bf[j][kk] = *(const bf16x8_t*)(tb + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
        }
    };
    auto quad_from = [&](int asub, int bsub, bf16x8_t (&a)[4][2], bf16x8_t (&bf)[2][2]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[asub * 4 + i][bsub * 2 + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][kk], bf[j][kk], acc[asub * 4 + i][bsub * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // the A half-tiles of K-tile kt+1 are requested in phase 0, the B half-tiles in phase 1, after that phase's fragment reads
    auto dma_phase = [&](int ph) {
      if (more) {
#pragma unroll
        for (int h = 2 * ph; h < 2 * ph + 2; ++h) {
          if (!last) dma_half(cur_t, h, kt + 1, nb);
          else {                                                  // the next item's offsets are computed here, once per item, and
            Tile nt;                                              // die at once: kept live through the K loop they are spilled
            make_tile(nxt.tile, nt);                              // (the kernel sits at the 256-register limit)
            dma_half(nt, h, nxt.k0, nb);
          }
        }
      }
    };
    // (Measured and removed, round 3: the eight LDS-DMA pieces of a wave SPREAD over the first two quads, one piece every four
    //  MFMAs, instead of two bursts of four in front of them -- 18-30 % SLOWER on every shape.  tools/gemm256_timeline.py: a
    //  piece costs its wave ~80-150 cycles of instruction issue wherever it stands (580-750 cycles for 12 fragment reads + 4
    //  pieces, ~300 for 8 + 4, against 1024 for the K-tile's 64 MFMAs); inside a quad those cycles come out of the wave's own
    //  MFMA stream, in front of it the partner wave's MFMAs cover most of them.  Likewise the SIMD partners' DMA bursts at
    //  different places: 4-13 % slower on the big-tile shapes.)
    // software-pipelined fragment reads: the fragments of phase p+1 are requested BEFORE the MFMAs of phase p (second A
    // fragment buffer, 224 of 256 registers), so only the first reads after the barrier expose their LDS latency
    read_b(0, bf0);
    read_a_to(0, af);
    dma_phase(0);
    read_b(1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    G_S(3);
    quad_from(0, 0, af, bf0);
    __builtin_amdgcn_sched_barrier(0);
    G_S(4);
    read_a_to(1, af1);
    dma_phase(1);
    __builtin_amdgcn_sched_barrier(0);
    G_S(5);
    quad_from(0, 1, af, bf1);
    __builtin_amdgcn_sched_barrier(0);
    G_S(6);
    quad_from(1, 1, af1, bf1);
    quad_from(1, 0, af1, bf0);
    G_S(7);
    G_FLUSH(kt - cur.k0);
  }

  __syncthreads();                                               // every wave is done with the K-tile buffer just consumed
  unsigned char* const cbuf = smem + ((gk - 1) & 1) * BUF_BYTES;  // (the other one is receiving the next item's first K-tile)
  bool finish = true;
  if (cur.k0 != 0 || cur.k1 != nk) {
    // ---- a cut tile: contributors are the positions pf .. pl whose ranges meet the tile's iterations [x0, x0 + nk)
    const int x0 = cur.tile * nk;
    auto pos_of = [&](int x) { return (int)((((unsigned)x + 1u) * (unsigned)G + (unsigned)I - 1u) / (unsigned)I) - 1; };   // largest q with sk_start(q) <= x
    const int pf = pos_of(x0), pl = pos_of(x0 + nk - 1), s = pl - pf + 1;
    int* const arrive = p.sk_flags + cur.tile;
    int* const done = p.sk_flags + P3V_SK_MAX_TILES + cur.tile;
    if (tid == 0) *(volatile int*)cbuf = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(*(volatile int*)cbuf);
    __syncthreads();                                             // (the epilogue stages through cbuf)
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_slabs, 0, 0xffffffff, 0x00020000);
    constexpr int SLAB = TM * TN * 4;
    if (ticket < s - 1) {
      // not the last to arrive: publish the accumulators (write-through: the reducer may sit on another XCD) and move on
      const int slot = 2 * pos + (x0 + cur.k0 == it_begin ? 0 : 1);
#pragma unroll
      for (int c = 0; c < 32; ++c)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[c >> 2][c & 3]), rs_s, tid * 16, slot * SLAB + c * 8192, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish = false;
    } else {
      // the last to arrive reduces.  The s - 1 others hold tickets, i.e. they are resident and storing: a bounded wait
      if (tid == 0) {
        for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < s - 1; ++spin)
          __builtin_amdgcn_s_sleep(4);
        __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      // Sum in POSITION order, whoever reduces: t = slab(pf) + ... + slab(own - 1) in registers, acc += t (= t + acc), then
      // acc += slab(own + 1) ... -- every update of the accumulators is an in-place add (a select between "own registers" and
      // "loaded slab" would give the array new registers and fill this path with copies and spills).
      const int first_slot = 2 * pf + (sk_start(pf) == x0 ? 0 : 1);   // pf's item opens the tile; is it pf's first item?
      const int j_own = pos - pf;
      auto slab_ld = [&](int j, int c) {
        const int slot = j == 0 ? first_slot : 2 * (pf + j);
        return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_s, tid * 16, slot * SLAB + c * 8192, 16));
      };
#pragma unroll
      for (int cg = 0; cg < 32; cg += 16) {
        if (j_own > 0) {
          f32x4_t t[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) t[e] = slab_ld(0, cg + e);
          for (int j = 1; j < j_own; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e] += slab_ld(j, cg + e);
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[(cg + e) >> 2][(cg + e) & 3] += t[e];
        }
        for (int j = j_own + 1; j < s; ++j) {
          f32x4_t v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = slab_ld(j, cg + e);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[(cg + e) >> 2][(cg + e) & 3] += v[e];
        }
      }
    }
  }

  if (finish) {
  // ---- epilogue: 128 x 64 per wave, in four passes of 32 rows through a wave-private [32][64] fp32 LDS tile that lives in
  // the K-tile buffer just consumed.  No padding fits in 8 KiB per wave: 16-byte column chunk c of row r sits at chunk
  // c ^ (r & 1) instead, which keeps the float4 read-back conflict-free.
  float* ct = (float*)cbuf + wave * (32 * 64);
  const int ccol = lane & 15, crow = (lane >> 4) * 4;
  auto ct_at = [&](int row, int col) { return ct + row * 64 + ((((col >> 2) ^ (row & 1)) << 2) | (col & 3)); };
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) *ct_at(i * 16 + crow + r, j * 16 + ccol) = acc[pass * 2 + i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (SILU) {
      const int c8 = (lane & 3) * 8, n = n0 + wc * 32 + c8;
#pragma unroll
      for (int it2 = 0; it2 < 2; ++it2) {
        const int row = it2 * 16 + (lane >> 2);
        const int m = m0 + wr * 128 + pass * 32 + row;
        if (m < p.M && n < p.N) {
          const float4 g0 = *(const float4*)ct_at(row, c8), g1 = *(const float4*)ct_at(row, c8 + 4);
          const float4 u0 = *(const float4*)ct_at(row, 32 + c8), u1 = *(const float4*)ct_at(row, 36 + c8);
          const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          const float us[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
          u32x4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float g = bf16_round(gs[2 * e + h]), u = bf16_round(us[2 * e + h]);
              o2[h] = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * u;
            }
            w[e] = pack_bf16x2(o2[0], o2[1]);
          }
          *(u32x4_t*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = w;
        }
      }
    } else {
      const int c8 = (lane & 7) * 8, n = n0 + wc * 64 + c8;
      const bool ncol_ok = n < p.N;
      float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ncol_ok && p.bias) {
        const u32x4_t bw = *(const u32x4_t*)(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) { bias[2 * j] = bf16lo(bw[j]); bias[2 * j + 1] = bf16hi(bw[j]); }
      }
#pragma unroll
      for (int it2 = 0; it2 < 4; ++it2) {
        const int row = it2 * 8 + (lane >> 3);
        const int m = m0 + wr * 128 + pass * 32 + row;
        if (m < p.M && ncol_ok) {
          const float4 a0 = *(const float4*)ct_at(row, c8), a1 = *(const float4*)ct_at(row, c8 + 4);
          float v[8] = {a0.x + bias[0], a0.y + bias[1], a0.z + bias[2], a0.w + bias[3],
                        a1.x + bias[4], a1.y + bias[5], a1.z + bias[6], a1.w + bias[7]};
          const size_t o = (size_t)m * p.ldo + n;
          if (EPI == P3V_EPI_BIAS_QGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));
          } else if (EPI == P3V_EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf2(v[e]);
          } else if (EPI == P3V_EPI_RESID_BF16) {
            const u32x4_t rw = *(const u32x4_t*)((const bf16_t*)p.resid + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo(rw[j]) + bf16_round(v[2 * j]); v[2 * j + 1] = bf16hi(rw[j]) + bf16_round(v[2 * j + 1]); }
          }
          if (EPI == P3V_EPI_BIAS_RESID_F32 || EPI == P3V_EPI_F32) {
            if (EPI == P3V_EPI_BIAS_RESID_F32) {
              const float4 r0 = *(const float4*)((const float*)p.resid + o), r1 = *(const float4*)((const float*)p.resid + o + 4);
              v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
            *(float4*)((float*)p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)p.out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            u32x4_t w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
            *(u32x4_t*)((bf16_t*)p.out + o) = w;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  }
  if (!has_next) break;
  cur = nxt;
  make_tile(cur.tile, cur_t);
  }                                                             // next item of this workgroup
}

// ---- launch plan: grid (a multiple of 8, at most one workgroup per CU) and the size of the shared (stream-K) region.
// T tiles on G workgroups.  T <= G: all of them shared when that fills the machine better than T whole tiles and a
// workgroup's share is still a real piece of work; T > G: the last G + T % G tiles are shared (every workgroup 1..2 tiles'
// worth, so no tile is cut more than once) unless the remainder is nearly a full round anyway.
struct Plan256 { int grid, sk_tiles; };
static int n_cu8() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;   // (no device: sizing only)
    n_cu = min(pr.multiProcessorCount / 8 * 8, P3V_SK_SLABS / 2);     // persistent grid: one workgroup per CU, a multiple of 8 (XCDs)
  }
  return n_cu;
}
static Plan256 plan256(int M, int N, int K, int epilogue, bool have_ws) {
  const int n_tile = epilogue == P3V_EPI_SILU_MUL ? TN / 2 : TN;
  const int T = p3v_cdiv(N, n_tile) * p3v_cdiv(M, TM), nk = K / TK, G = n_cu8();
  Plan256 pl = {min((T + 7) / 8 * 8, G), 0};
  const int mode = p3v_tuning().gemm_streamk;                     // -1 auto, 0 never, 1 wherever it is possible
  if (!have_ws || mode == 0 || T % G == 0) return pl;
  if (T < G) {
    const int per_wg = (int)((long)T * nk / G);                   // K-tiles per workgroup when shared
    if ((mode == 1 && per_wg >= 1) || (per_wg >= 8 && T * 10 <= G * 9 && T <= P3V_SK_MAX_TILES)) pl = {G, T};
  } else {
    const int R = T % G;
    if (mode == 1 || R * 100 <= G * 88) pl = {G, G + R};
  }
  return pl;
}

// cost of the plan in rounds of whole big tiles (p3v_gemm.hip prices it against its big / small row packings): the shared region
// costs its share of a round + the seam fix-up (a slab out, a slab in: ~6 us against 1.38 us per K-tile)
float p3v_gemm256_cost(int M, int N, int K, int epilogue, bool have_ws) {
  const int n_tile = epilogue == P3V_EPI_SILU_MUL ? TN / 2 : TN;
  const int T = p3v_cdiv(N, n_tile) * p3v_cdiv(M, TM), G = n_cu8();
  const Plan256 pl = plan256(M, N, K, epilogue, have_ws);
  if (!pl.sk_tiles) return (float)p3v_cdiv(T, G);
  return (float)T / G + 6.0f / (1.38f * (K / TK));
}

// true when the stream-K plan of this shape needs the caller's workspace (p3v_gemm_ws_bytes)
bool p3v_gemm256_wants_ws(int M, int N, int K, int epilogue) { return plan256(M, N, K, epilogue, true).sk_tiles > 0; }

template <int EPI>
static int launch_gemm256(Gemm256P& p, void* ws, int64_t ws_bytes, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm256<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const bool have_ws = ws && ws_bytes >= P3V_SK_WS_BYTES && !((uintptr_t)ws & 15);
  const Plan256 pl = plan256(p.M, p.N, p.K, EPI, have_ws);
  p.sk_tiles = pl.sk_tiles;
  p.sk_flags = (int*)ws;
  p.sk_slabs = (float*)((char*)ws + P3V_SK_HDR_BYTES);
  hipLaunchKernelGGL((k_gemm256<EPI>), dim3(pl.grid), dim3(512), GEMM256_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// called by p3v_gemm (which decides which problems get the big tile); returns P3V_ERR_UNSUPPORTED to fall back
int p3v_gemm256_try(const p3v_gemm_args_t* a, hipStream_t s) {
  const int n_tile = a->epilogue == P3V_EPI_SILU_MUL ? TN / 2 : TN;
  if (a->N % n_tile || a->K % TK || a->epilogue == P3V_EPI_PATCH) return P3V_ERR_UNSUPPORTED;
  const size_t w_rows = (size_t)a->N * (a->epilogue == P3V_EPI_SILU_MUL ? 2 : 1);
  if (w_rows * a->ldw * 2 >= ((size_t)1 << 32) || (size_t)256 * a->lda * 2 >= ((size_t)1 << 31)) return P3V_ERR_UNSUPPORTED;  // 32-bit buffer offsets
  Gemm256P p = {a->A, a->W, a->out, a->bias, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, 0, nullptr, nullptr};
  switch (a->epilogue) {
    case P3V_EPI_NONE: return launch_gemm256<P3V_EPI_NONE>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_BIAS: return launch_gemm256<P3V_EPI_BIAS>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_BIAS_QGELU: return launch_gemm256<P3V_EPI_BIAS_QGELU>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_BIAS_GELU: return launch_gemm256<P3V_EPI_BIAS_GELU>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_BIAS_RESID_F32: return launch_gemm256<P3V_EPI_BIAS_RESID_F32>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_RESID_BF16: return launch_gemm256<P3V_EPI_RESID_BF16>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_SILU_MUL: return launch_gemm256<P3V_EPI_SILU_MUL>(p, a->ws, a->ws_bytes, s);
    case P3V_EPI_F32: return launch_gemm256<P3V_EPI_F32>(p, a->ws, a->ws_bytes, s);
    default: return P3V_ERR_UNSUPPORTED;
  }
}
