// 256 x 256-tile GEMM, PING-PONG K loop (round 5).  Same tile, same LDS image, same fragments, same epilogue as
// p3v_gemm256.hip; what changes is WHEN each wave does what.
//
// There (rounds 2-4) all eight waves run the same sequence per K-tile -- barrier, fragment reads + LDS-DMA requests, 64 MFMAs
// with the rest of the reads in between -- so after every barrier the two waves of a SIMD are both in their load stretch
// (matrix pipe idle for ~600 cycles), then both want the matrix pipe; measured 0.70 of the pipe inside the loop
// (profiles/r03_gemm256_timeline.txt), and every attempt to shift the partners against each other inside that structure lost.
// Here the workgroup is two GROUPS of four waves, one wave of each on every SIMD (group g = wave >> 2 = the tile's upper / lower
// 128 rows), that alternate roles in lockstep, one s_barrier per phase:
//     phase A_s:  group 0: the 64 MFMAs of K-tile s, nothing else       group 1: all 24 fragment reads of K-tile s + its 8 DMA pieces
//     phase B_s:  group 1: the 64 MFMAs of K-tile s                     group 0: all 24 fragment reads of K-tile s+1 + its 8 DMA pieces
// so on every SIMD one wave issues back-to-back MFMAs out of registers (1024 cycles) while its partner does LDS / vector-memory
// work only (different issue ports), and the matrix pipe never waits for a fragment.  One fragment set (96 registers) is enough: a
// wave loads the next K-tile's fragments while it is not computing.
// LDS (all 160 KiB): the A halves are private to a group -> two 16-KiB slots each; the W tile is read by both groups half a K-tile
// apart -> THREE 32-KiB slots, so that every DMA request is issued at least two phases (one K-tile time, ~1 us) before its first
// read: group 0 requests, in phase B_s, its A half and the upper W half of K-tile s+2; group 1, in phase A_s, its A half of K-tile
// s+1 and the lower W half of K-tile s+2 (in the code: a group's load phase for step s requests step s+1 resp. s+2).  A wave waits for its own requests (vmcnt(0)) at the END of its compute phase -- they
// have had the whole phase to land -- and the phase barriers order them for the other waves; requests stay in flight across
// barriers (raw s_barrier, no vmcnt in front of it).
// The K-tile stream runs across output tiles (persistent workgroups, XCD-aware tile order as in p3v_gemm256.hip); each group
// stores its half of a finished tile straight from the accumulators in the phase after its last compute phase, i.e. while the
// other group computes.
#include <stdlib.h>

#include "p3v_gemm256_epi.h"

#define PP_A_SLOT HALF_BYTES                  // 16 KiB
#define PP_B_SLOT (2 * HALF_BYTES)            // 32 KiB
#define PP_A_BASE(g, slot) (((g) * 2 + (slot)) * PP_A_SLOT)                 // A0[2] A1[2]: 0 .. 64 KiB
#define PP_B_BASE(slot) (4 * PP_A_SLOT + (slot) * PP_B_SLOT)                // W[3]: 64 .. 160 KiB
#define GEMM256PP_LDS (4 * PP_A_SLOT + 3 * PP_B_SLOT)                       // 163840 B

struct TilePP { int m0, n0; __amdgpu_buffer_rsrc_t rs_a; };      // all wave-uniform (scalar registers)

template <int EPI>
__global__ void __launch_bounds__(512, 1) k_gemm256pp(Gemm256P p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: LDS-DMA bases stay in SGPRs
  const int grp = wave >> 2, wc = wave & 3;                       // group = row half of the tile (wr), wave column
  constexpr int n_out_tile = SILU ? TN / 2 : TN;
  const int gx = (p.N + n_out_tile - 1) / n_out_tile, gy = (p.M + TM - 1) / TM, nwg = gx * gy;
  const int nk = p.K / TK;                                        // >= 2 (launcher)
  const int G = gridDim.x;
  if ((int)blockIdx.x >= nwg) return;
  const int n_my = (nwg - 1 - (int)blockIdx.x) / G + 1;           // output tiles of this workgroup: blockIdx.x, + G, ...

  auto make_tile = [&](int wid, TilePP& t) {                      // scalar arithmetic only
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    constexpr int BAND = 4;
    const int band = id / (BAND * gx), in_band = id % (BAND * gx);
    const int rows = min(BAND, gy - band * BAND);
    const int m_t = band * BAND + in_band % rows, n_t = in_band / rows;
    t.m0 = m_t * TM, t.n0 = n_t * n_out_tile;
    t.rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)t.m0 * p.lda), 0, 0xffffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
  // DMA geometry: a group fills its A half (128 rows) and one W half (128 rows), 16 one-KiB pieces each = 4 per wave and half:
  // piece q of wave w4 = (wave & 3) covers half-tile rows q * 32 + w4 * 8 .. + 8; this lane: row + lane / 8, 16-byte chunk lane % 8,
  // fetched from the XOR-swizzled source chunk (the LDS image is lane-linear, p3v_gemm.hip).  The per-lane byte offsets are
  // RECOMPUTED at every request from an opaque copy of the lane id (~3 VALU operations each, in a phase that has the VALU to itself):
  // carried through the loop next to 128 accumulator and 96 fragment registers they are spilled, and a scratch reload in the
  // load phase waits vmcnt(0), i.e. for the DMA requests just issued.
  const int w4 = wave & 3;
  auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto dma_a = [&](const TilePP& t, int kt, int slot) {            // this group's A half of K-tile kt
    unsigned char* base = smem + PP_A_BASE(grp, slot) + w4 * 1024;
    const int l = opaque_lane(), srow = l >> 3, sw = ((l & 7) ^ (srow & 7)) * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = grp * 128 + q * 32 + w4 * 8 + srow;
      const int ar = min(t.m0 + rr, p.M - 1) - t.m0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rs_a, (lptr_t)(base + q * 4096), 16, (ar * p.lda + sw) * 2, kt * (TK * 2), 0, 0);
    }
  };
  auto dma_b = [&](const TilePP& t, int kt, int slot) {            // W half `grp` of K-tile kt
    unsigned char* base = smem + PP_B_BASE(slot) + grp * HALF_BYTES + w4 * 1024;
    const int l = opaque_lane(), srow = l >> 3, sw = ((l & 7) ^ (srow & 7)) * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = grp * 128 + q * 32 + w4 * 8 + srow;          // tile W row
      int br;
      if (SILU) {                                     // wave column group wcol (64 tile rows) = 32 gate + 32 up rows
        const int wcol = rr >> 6, ni = (rr & 63) >> 4, c = rr & 15;
        br = min(t.n0 + wcol * 32 + (ni & 1) * 16 + c, p.N - 1) + (ni >> 1) * p.N;
      } else {
        br = min(t.n0 + rr, p.N - 1);
      }
      const int vo = (int)(((unsigned)br * (unsigned)p.ldw + (unsigned)sw) * 2u);   // < 2^32: checked by the launcher
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(base + q * 4096), 16, vo, kt * (TK * 2), 0, 0);
    }
  };

  // ---- step bookkeeping: step s = j * nk + kt = K-tile kt of this workgroup's j-th output tile; A slot s & 1, W slot s % 3
  TilePP cur;                                                     // the tile this GROUP is computing (advanced after its epilogue)
  make_tile(blockIdx.x, cur);
  auto request = [&](bool next_tile, int j_next, int kt, bool a_half, int slot) {   // this wave's pieces of one half-tile
    if (!next_tile) {
      if (a_half) dma_a(cur, kt, slot); else dma_b(cur, kt, slot);
    } else {
      TilePP nt;
      make_tile((int)blockIdx.x + j_next * G, nt);
      if (a_half) dma_a(nt, kt, slot); else dma_b(nt, kt, slot);
    }
  };

  bf16x8_t af[8][2], bfr[4][2];
  auto read_frags = [&](int a_slot, int w_slot) {                 // all 24 fragments of a K-tile step
    const int l = opaque_lane(), frow = l & 15, fchunk = l >> 4;
    const unsigned char* ta = smem + PP_A_BASE(grp, a_slot);
    const unsigned char* tb = smem + PP_B_BASE(w_slot) + (wc >> 1) * HALF_BYTES + (wc & 1) * 64 * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r = j * 16 + frow;
        bfr[j][kk] = *(const bf16x8_t*)(tb + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r = i * 16 + frow;
        af[i][kk] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
  };
  f32x4_t acc[8][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  auto compute = [&]() {                                          // 64 MFMAs out of registers, then this wave's DMA must have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j][kk], af[i][kk], acc[i][j], 0, 0, 0);   // W first: transposed block
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                 // (raw: __syncthreads() would drain the DMA requests in flight)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto inc3 = [](int v) { return v == 2 ? 0 : v + 1; };

  // ---- Both groups run the SAME loop -- per step: load phase (fragments of step s, DMA requests), barrier, compute phase, barrier --
  // group 1 one phase behind group 0 (it idles through one extra barrier at the start, group 0 through one at the end), so a wave's
  // fragments live from its load phase to the compute phase right after it and nothing but the accumulators is carried around
  // the loop.  What the load phase of step s requests: its group's A half of step s + 1, and its W half of step s + 1 (group 0,
  // whose load phase is phase B_{s-1}) or s + 2 (group 1: phase A_s) -- in both cases the slot's last reader finished a whole
  // phase earlier and the first reader comes two phases later.
  const int w_ahead = grp + 1;
  zero_acc();
  request(false, 0, 0, true, 0);
  request(false, 0, 0, false, 0);
  if (grp == 1) request(false, 0, 1, false, 1);                   // (nk >= 2: step 1 is K-tile 1 of the first tile)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  phase_barrier();
  if (grp == 1) phase_barrier();

  int w_s = 0;                                                    // W slot of the current step (s % 3)
  for (int j = 0; j < n_my; ++j) {
    const bool more = j + 1 < n_my;
    if (j > 0) {                                                  // this group's half of tile j - 1, while the other group computes
      gemm256_epilogue<EPI>(p, acc, cur.m0, cur.n0, grp, wc, lane);
      zero_acc();
      make_tile((int)blockIdx.x + j * G, cur);
    }
    for (int kt = 0; kt < nk; ++kt) {
      const int a_s = (j * nk + kt) & 1;
      // ---- load phase of step s = j * nk + kt
      read_frags(a_s, w_s);
      if (kt + 1 < nk) request(false, 0, kt + 1, true, a_s ^ 1);
      else if (more) request(true, j + 1, 0, true, a_s ^ 1);
      const int kw = kt + w_ahead, w_slot = w_ahead == 1 ? inc3(w_s) : inc3(inc3(w_s));
      if (kw < nk) request(false, 0, kw, false, w_slot);
      else if (more) request(true, j + 1, kw - nk, false, w_slot);
      phase_barrier();
      // ---- compute phase
      compute();
      phase_barrier();
      w_s = inc3(w_s);
    }
  }
  gemm256_epilogue<EPI>(p, acc, cur.m0, cur.n0, grp, wc, lane);
  if (grp == 0) phase_barrier();
}

template <int EPI>
static int launch_gemm256pp(const Gemm256P& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm256pp<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256PP_LDS) != hipSuccess)
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
  hipLaunchKernelGGL((k_gemm256pp<EPI>), dim3(min(tiles, n_cu)), dim3(512), GEMM256PP_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// called by p3v_gemm256_try (p3v_gemm256.hip) when the ping-pong loop is selected; same contract
int p3v_gemm256pp_try(const p3v_gemm_args_t* a, hipStream_t s) {
  if (a->K < 2 * TK) return P3V_ERR_UNSUPPORTED;
  const Gemm256P p = {a->A, a->W, a->out, a->bias, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo};
  switch (a->epilogue) {
    case P3V_EPI_NONE: return launch_gemm256pp<P3V_EPI_NONE>(p, s);
    case P3V_EPI_BIAS: return launch_gemm256pp<P3V_EPI_BIAS>(p, s);
    case P3V_EPI_BIAS_QGELU: return launch_gemm256pp<P3V_EPI_BIAS_QGELU>(p, s);
    case P3V_EPI_BIAS_GELU: return launch_gemm256pp<P3V_EPI_BIAS_GELU>(p, s);
    case P3V_EPI_BIAS_RESID_F32: return launch_gemm256pp<P3V_EPI_BIAS_RESID_F32>(p, s);
    case P3V_EPI_RESID_BF16: return launch_gemm256pp<P3V_EPI_RESID_BF16>(p, s);
    case P3V_EPI_SILU_MUL: return launch_gemm256pp<P3V_EPI_SILU_MUL>(p, s);
    case P3V_EPI_F32: return launch_gemm256pp<P3V_EPI_F32>(p, s);
    default: return P3V_ERR_UNSUPPORTED;
  }
}
