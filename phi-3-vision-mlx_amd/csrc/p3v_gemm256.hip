// Large-tile variant of the dense projection for prefill-sized problems (N % 256 == 0, K % 64 == 0; p3v_gemm decides
// how many rows of a problem it gets):
//   256(M) x 256(N) x 64(K) tile, 512 threads = 8 waves as 2(M) x 4(N), each wave 128 x 64 =
//   8 x 4 v_mfma_f32_16x16x32_bf16 accumulators (128 registers), 2.67 MFMAs per LDS fragment read
//   (the 128x128 kernel: 2.0) and 25 % less global->LDS traffic per flop.
//   K-tile t lives in LDS buffer t&1 (2 x 64 KiB); a tile is four PHASES (one 64x32 accumulator quadrant = 16
//   MFMAs each).  The four 16-KiB half-tiles of tile t+1 are requested by LDS-DMA in the first two phases of
//   tile t, each batch AFTER that phase's fragment reads are issued (SCHED 0x50): measured against one
//   half-tile per phase (the last one then has only 16 MFMAs to land before the vmcnt(0)) +12 % at 4096^3
//   (1170 -> 1310 TF/s), against all four up front +7 % (eight DMA issues ahead of the first fragment reads
//   delay the first MFMA).  The fragment reads are software-pipelined (ORDER 2): the fragments of phase p+1 are requested
//   before the MFMAs of phase p (second A-fragment buffer, 256 registers, no spill), +1.5-2 % at 4096^3 / 8192^3
//   (1.28 -> 1.30, 1.32 -> 1.34 PFLOP/s).  One vmcnt(0) + barrier per K-tile.  No barrier inside a tile: the waves
//   de-phase, one wave's fragment reads overlap another's MFMAs.
//   Same XOR-swizzled LDS image and the same epilogues as p3v_gemm.hip; round 5: stored straight from the accumulators.
#include <stdlib.h>

#include "p3v_gemm256_epi.h"

#define BUF_BYTES (4 * HALF_BYTES)    // A0 A1 B0 B1
#define GEMM256_LDS (2 * BUF_BYTES)   // 128 KiB

struct Tile256 {
  int m0, n0; int a_off[2][2], b_off[2][2]; __amdgpu_buffer_rsrc_t rs_a;
  int swapped, is_k, row0;              // P3V_EPI_QKV: V tile (operand roles swapped) / K region / first W row of the region
};

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
template <int EPI, int SCHED = 0x50, int ORDER = 0>
__global__ void __launch_bounds__(512, 1) k_gemm256(Gemm256P p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL, QKV = EPI == P3V_EPI_QKV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: LDS-DMA bases stay in SGPRs
  const int wr = wave >> 2, wc = wave & 3;
  constexpr int n_out_tile = SILU ? TN / 2 : TN;
  const int gx = (p.N + n_out_tile - 1) / n_out_tile, gy = (p.M + TM - 1) / TM, nwg = gx * gy;
  // (P3V_EPI_QKV: N = (nh + 2 nkv) * hd; column tiles 0 .. nq-1 are Q, then K, then V -- p3v_gemm_qkv.h)
  const int nq_t = QKV ? p.q.nh * p.q.hd / TN : 0, nk_t = QKV ? p.q.nkv * p.q.hd / TN : 0;

  // ---- PERSISTENT over output tiles (round 3): workgroup b takes tiles b, b + G, b + 2G, ... (G = gridDim.x, a multiple of 8,
  // so a workgroup's tiles keep its XCD in the XCD-aware order below).  The K-tile stream runs ACROSS tile seams: during the
  // last K-tile of a tile the first K-tile of the workgroup's NEXT tile is requested into the other LDS buffer, and the
  // epilogue (staged through the buffer that was just consumed) runs under that DMA -- the 64-KiB cold fetch that opened every
  // tile (HBM latency + 4 x 16 KiB, ~2 us) was 15-20 % of a K = 1024 tile (the ViT projections) and 5 % at K = 3072.
  using Tile = Tile256;   // (declared outside the template: a builtin called with a member of a DEPENDENT type is only checked at
                          //  instantiation, fails there in the HOST pass -- no such builtin -- and hipcc silently drops the host stub)
  const int srow = tid >> 3, schunk = tid & 7;
  auto make_tile = [&](int wid, Tile& t) {
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    constexpr int BAND = 4;
    const int band = id / (BAND * gx), in_band = id % (BAND * gx);
    const int rows = min(BAND, gy - band * BAND);
    const int m_t = band * BAND + in_band % rows, n_t = in_band / rows;
    t.m0 = m_t * TM, t.n0 = n_t * n_out_tile;
    t.swapped = t.is_k = t.row0 = 0;
    if (QKV) {
      t.swapped = n_t >= nq_t + nk_t;
      t.is_k = !t.swapped && n_t >= nq_t;
      t.row0 = t.swapped ? (p.q.nh + p.q.nkv) * p.q.hd : t.is_k ? p.q.nh * p.q.hd : 0;
      t.n0 = (n_t - (t.swapped ? nq_t + nk_t : t.is_k ? nq_t : 0)) * TN;        // first column INSIDE the region
    }
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
        int br;
        if (SILU) {                                   // wave column group wcol (64 tile rows) = 32 gate + 32 up rows
          const int wcol = rr >> 6, ni = (rr & 63) >> 4, c = rr & 15;
          br = min(t.n0 + wcol * 32 + (ni & 1) * 16 + c, p.N - 1) + (ni >> 1) * p.N;
        } else if (QKV) {                             // Q / K: pair order (both halves of a rotation pair in one lane)
          br = qkv_pair_row(p.q, t.row0, t.n0 >> 1, rr);
        } else {
          br = min(t.n0 + rr, p.N - 1);
        }
        if (QKV && t.swapped) {
          // V tile: the tile's rows are the 256 W rows row0 + n0 .. (A side), its columns the tokens m0 .. (B side)
          t.a_off[h][qq] = (rr * p.ldw + sw) * 2;                                                  // from W + (row0 + n0) * ldw
          t.b_off[h][qq] = (int)(((unsigned)(t.m0 + ar) * (unsigned)p.lda + (unsigned)sw) * 2u);    // from A, clamped token row
        } else {
          t.a_off[h][qq] = (ar * p.lda + sw) * 2;
          t.b_off[h][qq] = (int)(((unsigned)br * (unsigned)p.ldw + (unsigned)sw) * 2u);   // < 2^32: checked by the launcher
        }
      }
    t.rs_a = QKV && t.swapped ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)(t.row0 + t.n0) * p.ldw), 0, 0xffffffff, 0x00020000)
                              : __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)t.m0 * p.lda), 0, 0xffffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffff, 0x00020000);   // (V tiles: tokens on the B side)
  auto dma_half = [&](const Tile& t, int which, int kt, int buf) {   // which: 0 A0, 1 A1, 2 B0, 3 B1
    unsigned char* base = smem + buf * BUF_BYTES + which * HALF_BYTES + wave * 1024;
    const int h = which & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const __amdgpu_buffer_rsrc_t rs = which < 2 ? t.rs_a : (QKV && t.swapped) ? rs_x : rs_w;
      const int vo = which < 2 ? t.a_off[h][q] : t.b_off[h][q];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(base + q * 8192), 16, vo, kt * (TK * 2), 0, 0);
    }
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  const int nk = p.K / TK;
  int wid = blockIdx.x;
  if (wid >= nwg) return;
  Tile cur_t;
  make_tile(wid, cur_t);
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) dma_half(cur_t, w4, 0, 0);
  int gk = 0;                                                     // K-tiles consumed so far: LDS buffer = gk & 1
  bool landed = false;                                            // this wave's pieces of the K-tile about to be consumed are already in LDS
  for (;;) {
  const int m0 = cur_t.m0, n0 = cur_t.n0;
  const bool has_next = wid + (int)gridDim.x < nwg;
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#ifdef P3V_G256_DEBUG
  unsigned long long g_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int kt = 0; kt < nk; ++kt, ++gk) {
    G_S(0);
    // (first K-tile of a later tile: waited for BEFORE the previous tile's stores were issued, below -- a vmcnt(0) here would
    //  wait for those stores to drain; so the barrier is the raw one, __syncthreads() would add that wait back)
    if (!(kt == 0 && landed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G_S(1);
    __builtin_amdgcn_s_barrier();
    G_S(2);
    const bool last = kt + 1 == nk;
    const bool more = !last || has_next;                          // something to prefetch: this tile's next K-tile, or the next tile's first
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
    auto read_a = [&](int sub) { read_a_to(sub, af); };
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
            acc[asub * 4 + i][bsub * 2 + j] =        // (W fragment first: the block comes out TRANSPOSED, see the epilogue)
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], a[i][kk], acc[asub * 4 + i][bsub * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    auto quad = [&](int asub, int bsub, bf16x8_t (&bf)[2][2]) { quad_from(asub, bsub, af, bf); };
    // half-tile h of K-tile kt+1 is requested in phase (SCHED >> 2h) & 3; ORDER 1 = after the phase's fragment reads
    auto dma_phase = [&](int ph) {
      if (more) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (((SCHED >> (2 * h)) & 3) == ph) {
            if (!last) dma_half(cur_t, h, kt + 1, nb);
            else {                                                // the next tile's offsets are computed here, once per tile, and
              Tile nt;                                            // die at once: kept live through the K loop they are spilled
              make_tile(wid + gridDim.x, nt);                     // (the kernel sits at the 256-register limit)
              dma_half(nt, h, 0, nb);
            }
          }
      }
    };
    // (Measured and removed, round 3: the eight LDS-DMA pieces of a wave SPREAD over the first two quads, one piece every four
    //  MFMAs, instead of two bursts of four in front of them -- 18-30 % SLOWER on every shape.  tools/gemm256_timeline.py: a
    //  piece costs its wave ~80-150 cycles of instruction issue wherever it stands (580-750 cycles for 12 fragment reads + 4
    //  pieces, ~300 for 8 + 4, against 1024 for the K-tile's 64 MFMAs); inside a quad those cycles come out of the wave's own
    //  MFMA stream, in front of it the partner wave's MFMAs cover most of them.)
    // (Likewise measured and removed: the SIMD partners' DMA bursts at different places -- waves 0-3 in front of quads 0 / 1,
    //  waves 4-7 behind them -- so that the two waves of a SIMD are not both in their "reads + 4 pieces" stretch right after
    //  the barrier: 4-13 % slower on the big-tile shapes.)
    if (ORDER == 2) {
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
      G_FLUSH(kt);
      continue;
    }
    // phase 0
    if (ORDER == 0) dma_phase(0);
    read_b(0, bf0);
    read_a(0);
    if (ORDER == 1) dma_phase(0);
    quad(0, 0, bf0);
    // phase 1
    if (ORDER == 0) dma_phase(1);
    read_b(1, bf1);
    if (ORDER == 1) dma_phase(1);
    quad(0, 1, bf1);
    // phase 2
    if (ORDER == 0) dma_phase(2);
    read_a(1);
    if (ORDER == 1) dma_phase(2);
    quad(1, 1, bf1);
    // phase 3
    dma_phase(3);
    quad(1, 0, bf0);
  }

  // The next tile's first K-tile was requested during the last K-tile above: wait for it NOW, ahead of the epilogue's stores.  The
  // vector-memory counter cannot tell loads from stores, so behind the stores the same wait would last until they have drained
  // (all 256 CUs store at once: the HBM write rate, 5-10 us per tile); this way the next tile's first 64 MFMAs run under the drain.
  if (has_next) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed = true;
  }
  if constexpr (QKV) {
    if (!cur_t.swapped) qkv_epilogue_rot<8>(p.q, acc, p.bias, cur_t.is_k, cur_t.row0, (n0 >> 1) + wc * 32, m0 + wr * 128, p.M, lane);
    else qkv_epilogue_vt<8>(p.q, acc, p.bias ? p.bias + cur_t.row0 : nullptr, n0 + wr * 128, m0 + wc * 64, p.M, lane);
  } else {
    gemm256_epilogue<EPI>(p, acc, m0, n0, wr, wc, lane);    // straight from the accumulators: p3v_gemm256_epi.h
  }
  if (!has_next) break;
  wid += gridDim.x;
  make_tile(wid, cur_t);
  }                                                             // next tile of this workgroup
}

template <int EPI, int SCHED, int ORDER>
static int launch_gemm256_v(const Gemm256P& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm256<EPI, SCHED, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const int n_tile = EPI == P3V_EPI_SILU_MUL ? TN / 2 : TN;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return P3V_ERR_HIP;
    n_cu = pr.multiProcessorCount / 8 * 8;                        // persistent grid: one workgroup per CU, a multiple of 8 (XCDs)
  }
  const int tiles = p3v_cdiv(p.N, n_tile) * p3v_cdiv(p.M, TM);
  const int persist = p3v_tuning().gemm_persistent;                // 0: one workgroup per tile (round 2), else the persistent loop
  dim3 grid(persist ? min(tiles, n_cu) : tiles);
  hipLaunchKernelGGL((k_gemm256<EPI, SCHED, ORDER>), grid, dim3(512), GEMM256_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <int EPI>
static int launch_gemm256(const Gemm256P& p, hipStream_t s) { return launch_gemm256_v<EPI, 0x50, 2>(p, s); }

// the qkv projection with split + RoPE + KV append in the epilogue (p3v_gemm_qkv, p3v_gemm.hip): rows of `a` on big tiles
int p3v_gemm256_qkv(const p3v_gemm_args_t* a, const QkvP& q, hipStream_t s) {
  if (a->N % TN || a->K % TK || (q.nh * q.hd) % TN || (q.nkv * q.hd) % TN) return P3V_ERR_UNSUPPORTED;
  if ((size_t)a->N * a->ldw * 2 >= ((size_t)1 << 32) || (size_t)a->M * a->lda * 2 >= ((size_t)1 << 32)) return P3V_ERR_UNSUPPORTED;
  const Gemm256P p = {a->A, a->W, nullptr, a->bias, nullptr, a->M, a->N, a->K, a->lda, a->ldw, 0, q};
  return launch_gemm256<P3V_EPI_QKV>(p, s);
}

// called by p3v_gemm (which decides how many rows get the big tile); returns P3V_ERR_UNSUPPORTED to fall back
int p3v_gemm256_try(const p3v_gemm_args_t* a, hipStream_t s) {
  const int n_tile = a->epilogue == P3V_EPI_SILU_MUL ? TN / 2 : TN;
  if (a->N % n_tile || a->K % TK || a->epilogue == P3V_EPI_PATCH) return P3V_ERR_UNSUPPORTED;
  const size_t w_rows = (size_t)a->N * (a->epilogue == P3V_EPI_SILU_MUL ? 2 : 1);
  if (w_rows * a->ldw * 2 >= ((size_t)1 << 32) || (size_t)256 * a->lda * 2 >= ((size_t)1 << 31)) return P3V_ERR_UNSUPPORTED;  // 32-bit buffer offsets
  const Gemm256P p = {a->A, a->W, a->out, a->bias, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, {}};
  switch (a->epilogue) {
    case P3V_EPI_NONE: return launch_gemm256<P3V_EPI_NONE>(p, s);
    case P3V_EPI_BIAS: return launch_gemm256<P3V_EPI_BIAS>(p, s);
    case P3V_EPI_BIAS_QGELU: return launch_gemm256<P3V_EPI_BIAS_QGELU>(p, s);
    case P3V_EPI_BIAS_GELU: return launch_gemm256<P3V_EPI_BIAS_GELU>(p, s);
    case P3V_EPI_BIAS_RESID_F32: return launch_gemm256<P3V_EPI_BIAS_RESID_F32>(p, s);
    case P3V_EPI_RESID_BF16: return launch_gemm256<P3V_EPI_RESID_BF16>(p, s);
    case P3V_EPI_SILU_MUL: return launch_gemm256<P3V_EPI_SILU_MUL>(p, s);
    case P3V_EPI_F32: return launch_gemm256<P3V_EPI_F32>(p, s);
    default: return P3V_ERR_UNSUPPORTED;
  }
}
