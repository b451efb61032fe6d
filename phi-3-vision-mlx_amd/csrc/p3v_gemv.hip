// Skinny projection for decode:  y[M,N] = x[M,K] * W[N,K]^T,  M <= 8.
//
// Pure weight streaming (HBM-bound: 2*N*K bytes of W per call, read once):
//   * x (optionally RMS-normalised on the fly -- the pre-norm of the decoder
//     layer is fused here, phi.py:482,484) is staged once per block in LDS as
//     bf16, M*K*2 bytes;
//   * each wave owns output rows two at a time (for SiLU*up the pair is
//     gate row n / up row n+N, phi.py:469-471), lanes stride over K in 16-byte
//     chunks (8 bf16), so every global load is a fully coalesced 1 KiB
//     wave-load straight to VGPRs (no LDS round trip for W: it is used once);
//   * fp32 accumulate, 64-lane shuffle reduction, fused epilogue.
// Algorithmic bytes per launch: 2*N*K (+ M*K*2 for x per block from L2).
#include <stdlib.h>

#include <type_traits>

#include "p3v_common.h"
#include "p3v_gemv3_body.h"


template <int MT>
__global__ void __launch_bounds__(256) k_gemv(GemvP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[16];
  u32x4_t* xs = (u32x4_t*)smem;                       // [MT][chunks]
  const int chunks = p.K >> 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- stage x (+ fused RMSNorm) into LDS
  for (int m = 0; m < MT; ++m) {
    const bool live = m < p.M;
    const u32x4_t* xr = (const u32x4_t*)(p.x + (size_t)(live ? m : 0) * p.K);
    if (p.norm_w) {
      float ss = 0.f;
      for (int c = tid; c < chunks; c += 256) {
        const u32x4_t v = xr[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf16lo(v[j]), b = bf16hi(v[j]); ss += a * a + b * b; }
      }
      const float r = rsqrtf(block_sum(ss, red) / (float)p.K + p.eps);
      const u32x4_t* g = (const u32x4_t*)p.norm_w;
      for (int c = tid; c < chunks; c += 256) {
        const u32x4_t v = xr[c], gw = g[c];
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = rms_pair(v[j], r, gw[j]);
        xs[m * chunks + c] = live ? o : (u32x4_t){0, 0, 0, 0};
      }
    } else {
      for (int c = tid; c < chunks; c += 256) xs[m * chunks + c] = live ? xr[c] : (u32x4_t){0, 0, 0, 0};
    }
  }
  __syncthreads();

  const bool silu = p.epi == P3V_EPI_SILU_MUL;
  const int n_waves = gridDim.x * 4;
  for (int u = blockIdx.x * 4 + wave; u < p.units; u += n_waves) {
    const int r0 = silu ? u : 2 * u;
    int r1 = silu ? u + p.N : 2 * u + 1;
    const bool has1 = silu || r1 < p.N;
    if (!has1) r1 = r0;
    const u32x4_t* w0 = (const u32x4_t*)(p.W + (size_t)r0 * p.K);
    const u32x4_t* w1 = (const u32x4_t*)(p.W + (size_t)r1 * p.K);
    float a0[MT], a1[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a0[m] = a1[m] = 0.f;
    int c = lane;
    for (; c + 64 < chunks; c += 128) {             // two chunks per row in flight per iteration
      const u32x4_t wa = w0[c], wb = w1[c], wc = w0[c + 64], wd = w1[c + 64];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x4_t xa = xs[m * chunks + c], xb = xs[m * chunks + c + 64];
        a0[m] = dot8(wa, xa, a0[m]);
        a1[m] = dot8(wb, xa, a1[m]);
        a0[m] = dot8(wc, xb, a0[m]);
        a1[m] = dot8(wd, xb, a1[m]);
      }
    }
    for (; c < chunks; c += 64) {
      const u32x4_t wa = w0[c], wb = w1[c];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x4_t xa = xs[m * chunks + c];
        a0[m] = dot8(wa, xa, a0[m]);
        a1[m] = dot8(wb, xa, a1[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) { a0[m] = wave_sum(a0[m]); a1[m] = wave_sum(a1[m]); }
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m >= p.M) break;
        if (silu) {
          const float g = bf16_round(a0[m]), up = bf16_round(a1[m]);
          const float s = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g))));
          ((bf16_t*)p.out)[(size_t)m * p.N + u] = f32_to_bf16(s * up);
        } else {
          for (int h = 0; h < (has1 ? 2 : 1); ++h) {
            const float v = h ? a1[m] : a0[m];
            const size_t o = (size_t)m * p.N + (h ? r1 : r0);
            if (p.epi == P3V_EPI_F32) ((float*)p.out)[o] = v;
            else if (p.epi == P3V_EPI_RESID_BF16)
              ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32(p.resid[o]) + bf16_round(v));
            else ((bf16_t*)p.out)[o] = f32_to_bf16(v);
          }
        }
      }
    }
  }
}

template <int MT>
static int launch_gemv(const GemvP& p, hipStream_t s) {
  const size_t lds = (size_t)MT * p.K * 2;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemv<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  int blocks = p3v_cdiv(p.units, 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_gemv<MT>, dim3(blocks), dim3(256), lds, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------------------
// Streaming variant for the B=1 decode shapes (M <= 2; K = 3072 or 8192, compile-time):
// written so that hipcc can keep COUNTED s_waitcnt vmcnt(N) everywhere -- every load is
// unconditional and the pipeline body is branch-free (a predicated load or a branch between
// issue and use makes the compiler fall back to vmcnt(0), which drains the prefetch):
//   * x (+ norm weight) chunks are requested first, then the wave's first weight stage, so
//     the RMSNorm prologue waits only for the older x loads while the weights stream in;
//   * (row pair, K stage) software pipeline with two register buffers: stage s+1 is in
//     flight (2*CH 16-byte loads per lane) while stage s is reduced; the residual needed by
//     the epilogue travels with the stage (no dependent load at the end of a row);
//   * grid = ~8 waves per CU, each wave owning a contiguous run of row pairs (a pure
//     streaming read on this chip peaks at 2 blocks x 256 threads per CU, see tools/stream_floor.hip).
#ifdef P3V_GEMV_TIMING                                         // tools/gemv_timeline.py: 100 MHz stamps per wave (entry, exit)
__device__ long long p3v_gemv_tbuf[4096 * 2];
extern "C" int p3v_gemv_timing_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_gemv_tbuf), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#define GMARK(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) p3v_gemv_tbuf[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (k)] = wall_clock64(); } while (0)
#else
#define GMARK(k)
#endif
template <int MT, int NST, int CH>
__global__ void __launch_bounds__(256) k_gemv3(GemvP p, int units_per_wave, int wpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8];
  GMARK(0);
  gemv3_body<MT, NST, CH>(p, units_per_wave, blockIdx.x, smem, red, wpw);
  GMARK(1);
}

template <int MT, int NST, int CH, int STEP>
__global__ void __launch_bounds__(256) k_gemv3_step(GemvP p, int units_per_wave, int wpw, GemvStepP sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8];
  gemv3_body<MT, NST, CH, STEP>(p, units_per_wave, blockIdx.x, smem, red, wpw, &sp);
}

template <int MT, int NST, int CH, int STEP = STEP_NONE>
static int launch_gemv3(const GemvP& p, hipStream_t s, const GemvStepP* sp = nullptr) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return P3V_ERR_HIP;
    n_cu = pr.multiProcessorCount;
  }
  const int wpc = p3v_tuning().gemv_wpc;    // waves per CU
  int upw = p3v_cdiv(p.units, n_cu * wpc);               // row pairs per wave
  if (upw < 1) upw = 1;
  const int waves = p3v_cdiv(p.units, upw);
  const int wpw_ = p3v_gemv_wpw(waves, n_cu, p3v_tuning().gemv_wpw);   // 4 or 3 row-streaming waves per workgroup
  const size_t lds = (size_t)MT * p.K * 2;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k_gemv3<MT, NST, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  if constexpr (STEP != STEP_NONE) {
    if (p3v_cdiv(waves, wpw_) > P3V_GEMV_STEP_MAX_WG) return P3V_ERR_UNSUPPORTED;      // (amax_ws holds one candidate per workgroup and row)
    hipLaunchKernelGGL((k_gemv3_step<MT, NST, CH, STEP>), dim3(p3v_cdiv(waves, wpw_)), dim3(256), lds, s, p, upw, wpw_, *sp);
  } else {
    hipLaunchKernelGGL((k_gemv3<MT, NST, CH>), dim3(p3v_cdiv(waves, wpw_)), dim3(256), lds, s, p, upw, wpw_);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------------------
// Batched decode / constrained decoding (2 <= M <= 16 rows of x): the same weight stream, but the
// dot products go through the matrix cores so that the cost does not grow with M:
//   C[n, m] = W[n, :] . x[m, :]   v_mfma_f32_16x16x32_bf16 with A = 16 weight rows loaded STRAIGHT from
//   HBM into the fragment (lane l: row l&15, 16 bytes at k = 32*ks + 8*(l>>4); two consecutive k-steps
//   complete each 128-byte line) and B = x^T loaded straight from L2 in the same shape -- no LDS staging,
//   no barriers in the stream.  A workgroup = 16 output rows (SiLU*up: 16 gate + the matching 16 up rows);
//   its 4 waves split K four ways and merge their 16x16 partials (and the RMSNorm sums of squares, which
//   ride along with the x fragments) through LDS once at the end.
// The fused RMSNorm is applied as  (W . bf16(x*g)) * rsqrt(mean x^2 + eps)  -- the row factor commutes
// with the dot product, so no pass over x is needed before streaming starts.
#define GM_G 8                           // k-steps per pipeline stage (8 x 32 = 256 k)

template <bool SILU>
__global__ void __launch_bounds__(256) k_gemv_mfma(GemvP p) {
  __shared__ float cpart[4][2][256];
  __shared__ float sspart[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int n_base = blockIdx.x * 16;
  const int kq = p.K >> 2, k_lo = wave * kq, n_st = kq / (32 * GM_G);          // this wave's K quarter
  const int n_row = min(n_base + li, p.N - 1);                                  // clamped: loads stay unconditional
  const bf16_t* w0 = p.W + (size_t)n_row * p.K + k_lo + 8 * g;
  const bf16_t* w1 = p.W + (size_t)(n_row + p.N) * p.K + k_lo + 8 * g;          // up rows (SILU only)
  const bf16_t* xp = p.x + (size_t)min(li, p.M - 1) * p.K + k_lo + 8 * g;
  const bf16_t* gp = p.norm_w ? p.norm_w + k_lo + 8 * g : nullptr;

  u32x4_t wa[2][GM_G], wb[2][GM_G], xa[2][GM_G], ga[2][GM_G];
  auto issue = [&](int st, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int ks = 0; ks < GM_G; ++ks) {
      const int off = (st * GM_G + ks) * 32;
      wa[buf][ks] = __builtin_nontemporal_load((const u32x4_t*)(w0 + off));
      if (SILU) wb[buf][ks] = __builtin_nontemporal_load((const u32x4_t*)(w1 + off));
      xa[buf][ks] = *(const u32x4_t*)(xp + off);
      if (gp) ga[buf][ks] = *(const u32x4_t*)(gp + off);
    }
  };
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
  auto compute = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int ks = 0; ks < GM_G; ++ks) {
      u32x4_t xv = xa[buf][ks];
      if (gp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf16lo(xv[j]), b = bf16hi(xv[j]);
          ss += a * a + b * b;
          xv[j] = pack_bf16x2(a * bf16lo(ga[buf][ks][j]), b * bf16hi(ga[buf][ks][j]));
        }
      }
      const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xv);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[buf][ks]), xb, acc0, 0, 0, 0);
      if (SILU) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[buf][ks]), xb, acc1, 0, 0, 0);
    }
  };
  issue(0, IC0{});
  int st = 0;
  while (st + 2 < n_st) {                                             // branch-free body: counted vmcnt survives
    issue(st + 1, IC1{}); compute(IC0{});
    issue(st + 2, IC0{}); compute(IC1{});
    st += 2;
  }
  if (st + 1 < n_st) { issue(st + 1, IC1{}); compute(IC0{}); compute(IC1{}); }
  else compute(IC0{});

  // ---- merge the four K quarters.  C layout: lane holds outputs n = n_base + 4*g + r of x row m = li
  ss = rows_sum(ss);
  if (g == 0) sspart[wave][li] = ss;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    cpart[wave][0][lane * 4 + r] = acc0[r];
    if (SILU) cpart[wave][1][lane * 4 + r] = acc1[r];
  }
  __syncthreads();
  if (wave != 0 || li >= p.M) return;
  const float rs = gp ? rsqrtf(((sspart[0][li] + sspart[1][li]) + (sspart[2][li] + sspart[3][li])) / (float)p.K + p.eps) : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n_base + 4 * g + r, e = lane * 4 + r;
    if (n >= p.N) continue;
    const float v0 = ((cpart[0][0][e] + cpart[1][0][e]) + (cpart[2][0][e] + cpart[3][0][e])) * rs;
    const size_t o = (size_t)li * p.N + n;
    if (SILU) {
      const float v1 = ((cpart[0][1][e] + cpart[1][1][e]) + (cpart[2][1][e] + cpart[3][1][e])) * rs;
      const float gt = bf16_round(v0), up = bf16_round(v1);
      ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
    } else if (p.epi == P3V_EPI_F32) {
      ((float*)p.out)[o] = v0;
    } else if (p.epi == P3V_EPI_RESID_BF16) {
      ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32(p.resid[o]) + bf16_round(v0));
    } else {
      ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
    }
  }
}

// ---------------------------------------------------------------------------
// 5 <= M <= 8 rows (batched decode, B = 8 per GPU).  Three things the k_gemv_mfma stream above gets wrong at M = 8
// (qkv, 56.6 MB: 21.7 us = 2.6 TB/s) and this kernel fixes:
//   * WHOLE cache lines per load instruction.  An MFMA A fragment loaded straight from memory (lane -> row lane&15,
//     16 bytes at k-group lane>>4) touches 16 rows x 64 B = sixteen HALF lines per instruction; a pure stream with that
//     pattern tops out at 3.98 TB/s, one with eight full 128-byte lines per instruction at 5.5-5.6
//     (tools/scratch/frag_stream.hip: patterns A / C, F).  So the 16 MFMA rows here are 8 weight rows x 2 k-halves:
//     MFMA row i = 2*r + h, lane (i, g) loads row r, 16-byte chunk 2g + h of a 64-element line -- lane pairs are 32
//     contiguous bytes, the instruction covers 8 full lines (pattern F: 5.54 TB/s).  The B operand is laid out the same
//     way (column j = 2*m + h: x row m, chunk 2g + h), so C[2r + h][2m + h'] is the (row r, x row m) dot product over the
//     k-half h when h == h' and junk otherwise; the two halves are added with one DPP lane swap at the end.  Half the
//     matrix-core work is thrown away -- it is idle anyway.
//   * the activations are staged ONCE per workgroup in LDS as bf16(x * r * g) -- exactly the M = 1 kernel's RMSNorm
//     arithmetic -- instead of two more L2 loads (x, norm weights) per lane next to every weight load.
//   * a wave's whole weight slice (NST stages of 8 loads; 3 for K = 3072) is requested up front: the x loads go first
//     (loads retire in order and these come back from L2), then two weight stages, the third as soon as the raw x is
//     parked in LDS.  Every workgroup of a launch is resident at once, so with a two-deep pipeline the chip moved in
//     lockstep through [stage 0 in flight] [prologue] [stage 1 in flight] ...
// A workgroup = 16 weight rows (two 8-row sets; SiLU*up: 8 gate rows + the matching 8 up rows); wave w owns K slice w and
// stages that slice of x itself; the only workgroup barrier before the stream is the RMSNorm sum-of-squares exchange.
// LDS row stride 2K + 64 bytes: the 16 (x row, chunk) slots of every ds_read_b128 lane group fall on 16 different
// 16-byte bank groups.  K = 3072: 4 waves, 49.7 KB -> 3 workgroups / CU.  K = 8192: 8 waves, 131 KB -> 1 workgroup / CU.
#ifdef P3V_ATTN_TIMING                                         // tools/gemv8_timeline.py: 100 MHz timestamps per workgroup
__device__ long long p3v_gbuf[4096 * 8];
#define GMARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) p3v_gbuf[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
extern "C" int p3v_gemv_timing_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_gbuf), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#else
#define GMARK(k)
#endif
// Round 2: PERSISTENT over row sets.  The per-workgroup timeline (-DP3V_ATTN_TIMING, tools/gemv8_timeline.py) of the
// one-set-per-workgroup form (qkv: 576 workgroups, 3 per CU) showed the activations arriving 2-11 us after entry (27.6 MB
// of x reads of the same 48 KB by 2304 waves) and the RMSNorm pass over 8 x 768 values per wave costing another 3 us with
// 12 waves per CU doing it at once: the weights had landed long before anybody could use them (16.2 us for 56.6 MB).  Now
// a workgroup stages x ONCE and walks `sets` strided row sets (grid <= 512), a wave's whole slice of the NEXT set
// being requested stage by stage as the stages of the current one are consumed; the K-slice partials of a set are
// exchanged through a double-buffered 2 KB LDS block (one barrier per set).  The RMSNorm scale r of a row is a scalar: it
// is applied to the dot products in the epilogue (as k_gemv_mfma above does), the matrix cores see bf16(x * g) -- one bf16
// rounding of the activation, as in the reference, but no exchange of sums and no second pass over the slice before the
// first MFMA (that pass took 3 us; the sums are exchanged by the barrier the partials need anyway).  (Two slices in
// flight per wave -- tried: the 32 extra load instructions block the wave in issue while the queues are full and the
// prologue behind them gets later, not earlier.)
template <bool SILU, int NW, int NST>
__global__ void __launch_bounds__(NW * 64) k_gemv_mfma8(GemvP p, int n_sets) {
  constexpr int KQ = NST * 256, K = KQ * NW, XS = K * 2 + 64, NCH = KQ / 8;     // slice elements, LDS row stride (bytes), slice chunks
  constexpr int ROWS = SILU ? 8 : 16;                                            // output columns per set
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[NW][8];
  __shared__ float cpart[2][NW * 2 * 8 * 8];                                     // [parity][wave][row set][weight row][x row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int r8 = (lane & 15) >> 1, chunk = 2 * g + (lane & 1);                   // weight row in the set / x row; 16-byte chunk of a line
  const int k_lo = wave * KQ;
  auto row_ptrs = [&](int set, const bf16_t*& w0, const bf16_t*& w1) {
    const int n_base = set * ROWS;
    const int row0 = min(n_base + r8, p.N - 1);
    const int row1 = SILU ? p.N + row0 : min(n_base + 8 + r8, p.N - 1);
    w0 = p.W + (size_t)row0 * K + k_lo + 8 * chunk;
    w1 = p.W + (size_t)row1 * K + k_lo + 8 * chunk;
  };
  int set = blockIdx.x;
  const bf16_t *w0, *w1;
  row_ptrs(set, w0, w1);

  u32x4_t wa[1][NST][8];                                                         // stage = 4 lines x 2 row sets; one whole slice
  auto issue = [&](const bf16_t* a0, const bf16_t* a1, auto stc, auto slotc) {
    constexpr int st = decltype(stc)::value, slot = decltype(slotc)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wa[slot][st][2 * q] = __builtin_nontemporal_load((const u32x4_t*)(a0 + (st * 4 + q) * 64));
      wa[slot][st][2 * q + 1] = __builtin_nontemporal_load((const u32x4_t*)(a1 + (st * 4 + q) * 64));
    }
  };
  typedef std::integral_constant<int, 2> IC2;
  typedef std::integral_constant<int, 3> IC3;

  // ---- x slice (8 rows x NCH chunks, lanes over chunks) first, then the weight stages of the first set
  GMARK(0);
  unsigned char* xslice = smem + k_lo * 2;
  u32x4_t gv[2];
  float ss[8];
  {
    u32x4_t xv[8][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = min(lane + 64 * k, NCH - 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) xv[m][k] = *(const u32x4_t*)(p.x + (size_t)min(m, p.M - 1) * K + k_lo + 8 * c);
      gv[k] = p.norm_w ? *(const u32x4_t*)(p.norm_w + k_lo + 8 * c) : (u32x4_t){0, 0, 0, 0};
    }
    issue(w0, w1, IC0{}, IC0{});
    if constexpr (NST > 1) issue(w0, w1, IC1{}, IC0{});
    if (p.norm_w) {                                                              // RMSNorm: park bf16(x * g), keep the sums of squares
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        ss[m] = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int c = lane + 64 * k;
          if (c < NCH) {
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = bf16lo(xv[m][k][j]), b = bf16hi(xv[m][k][j]);
              ss[m] += a * a + b * b;
              o[j] = pack_bf16x2(a * bf16lo(gv[k][j]), b * bf16hi(gv[k][j]));
            }
            *(u32x4_t*)(xslice + m * XS + c * 16) = m < p.M ? o : (u32x4_t){0, 0, 0, 0};
          }
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int c = lane + 64 * k;
          if (c < NCH) *(u32x4_t*)(xslice + m * XS + c * 16) = m < p.M ? xv[m][k] : (u32x4_t){0, 0, 0, 0};
        }
    }
  }
  GMARK(1);
  if constexpr (NST > 2) issue(w0, w1, IC2{}, IC0{});
  if constexpr (NST > 3) issue(w0, w1, IC3{}, IC0{});
  GMARK(2);
  if (p.norm_w) {                                                                // sums of squares: read after the first set's barrier
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float t = wave_sum(ss[m]);
      if (lane == 0) red[wave][m] = t;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // the slice is wave-private: no barrier
  GMARK(3);

  const unsigned char* xrow = xslice + r8 * XS + chunk * 16;
  const bool odd = lane & 1;
  int par = 0;
  // one row set out of register slot `slot`; its registers are refilled stage by stage with the set one stride ahead
  // Two straight-line copies of the set body: REFILL (a further set follows: its stages are requested as this one's are consumed) and the
  // last set (nothing requested).  One body with `if (has_next) issue(..)` makes the compiler count vmcnt as if the refills did not
  // exist: the last stage of every set then waited for the refills issued a moment before (vmcnt(0)), one memory round trip per set.
  // Same reason for fetching the epilogue's residual element at the top, ahead of the refills, instead of in the epilogue (where it
  // would be the youngest load in flight).
  auto do_set = [&](auto slotc, auto refillc) {
    constexpr int slot = decltype(slotc)::value;
    constexpr bool REFILL = decltype(refillc)::value;
    const int nset = set + (int)gridDim.x;
    const bf16_t *w0n = nullptr, *w1n = nullptr;
    if constexpr (REFILL) row_ptrs(nset, w0n, w1n);
    const int e_R = tid & 7, e_m = (tid >> 3) & 7, e_sub = tid >> 6;
    const int e_n = set * ROWS + e_sub * 8 + e_R;
    const bool e_live = tid < (SILU ? 64 : 128) && e_m < p.M && e_n < p.N;
    const size_t e_o = (size_t)e_m * p.N + e_n;
    const bool e_has = !SILU && p.epi == P3V_EPI_RESID_BF16;
    uint32_t e_res = (e_has ? p.resid : p.x)[e_has && e_live ? e_o : 0];
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st < NST) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x8_t xb = *(const bf16x8_t*)(xrow + (st * 4 + q) * 128);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[slot][st][2 * q]), xb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa[slot][st][2 * q + 1]), xb, acc1, 0, 0, 0);
        }
        if constexpr (REFILL) { issue(w0n, w1n, stc, slotc); __builtin_amdgcn_sched_barrier(0); }   // (pinned behind this stage's MFMAs)
      }
    };
    step(IC0{}); step(IC1{}); step(IC2{}); step(IC3{});
    static_assert(NST <= 4, "unrolled for at most four stages");

    // ---- C[4*(lane>>4) + e][lane&15]: with column j = 2m + h only elements e = h (weight row 2*(lane>>4)) and e = h + 2
    // (row 2*(lane>>4) + 1) are dot products; add the two k-halves (lanes j = 2m, 2m + 1), then the NW K slices through LDS
    float e[2][2] = {{odd ? acc0[1] : acc0[0], odd ? acc0[3] : acc0[2]}, {odd ? acc1[1] : acc1[0], odd ? acc1[3] : acc1[2]}};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) e[t][u] += P3V_DPP_F32(e[t][u], 0xB1);        // quad_perm [1,0,3,2]
    GMARK(4);
    float* cp = cpart[par];
    if (!odd) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) cp[((wave * 2 + t) * 8 + 2 * g + u) * 8 + r8] = e[t][u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                // (the weight loads of the sets ahead stay in flight across it)
    GMARK(6);
    if (e_live) {
      const int R = e_R, m = e_m, sub = e_sub;
      {
        float v0 = 0.f, v1 = 0.f, t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          v0 += cp[((w * 2 + sub) * 8 + R) * 8 + m];
          if (SILU) v1 += cp[((w * 2 + 1) * 8 + R) * 8 + m];
          t += red[w][m];
        }
        if (p.norm_w) {
          const float r = rsqrtf(t / (float)K + p.eps);
          v0 *= r;
          v1 *= r;
        }
        const size_t o = e_o;
        if (SILU) {
          const float gt = bf16_round(v0), up = bf16_round(v1);
          ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
        } else if (p.epi == P3V_EPI_F32) {
          ((float*)p.out)[o] = v0;
        } else if (p.epi == P3V_EPI_RESID_BF16) {
          asm volatile("" : "+v"(e_res));                                        // (first looked at here: no wait for it above)
          ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32((bf16_t)e_res) + bf16_round(v0));
        } else {
          ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
        }
      }
    }
    set += (int)gridDim.x;
    par ^= 1;
  };
  while (set + (int)gridDim.x < n_sets) do_set(IC0{}, std::true_type{});
  do_set(IC0{}, std::false_type{});
}

template <bool SILU, int NW, int NST>
static int launch_gemv_mfma8(const GemvP& p, hipStream_t s) {
  const size_t lds = (size_t)8 * (p.K * 2 + 64);
  static bool attr_set = false;
  if (!attr_set && lds > 40 * 1024) {
    if (hipFuncSetAttribute((const void*)k_gemv_mfma8<SILU, NW, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8704) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  // <= 256 workgroups (one per CU: 192-256 VGPRs per wave), each walking `per` strided row sets
  // (576 sets -> 192 x 3, 1024 -> 256 x 4, 2004 -> 251 x 8, 192 -> 192 x 1)
  const int max_wg = p3v_tuning().gemv8_wgs;
  const int n_sets = p3v_cdiv(p.N, SILU ? 8 : 16), per = p3v_cdiv(n_sets, max_wg), grid = p3v_cdiv(n_sets, per);
  hipLaunchKernelGGL((k_gemv_mfma8<SILU, NW, NST>), dim3(grid), dim3(NW * 64), lds, s, p, n_sets);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}


static int launch_gemv_mfma(const GemvP& p, hipStream_t s) {
  dim3 grid(p3v_cdiv(p.N, 16));
  if (p.epi == P3V_EPI_SILU_MUL) hipLaunchKernelGGL(k_gemv_mfma<true>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_gemv_mfma<false>, grid, dim3(256), 0, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// The first / last projection of a replayed greedy step with p3v_step_begin / p3v_step_end folded in (gemv3_body<.., STEP>): only on
// the M = 1 streaming kernel (bf16 weights, K = 3072 or 8192); anything else reports P3V_ERR_UNSUPPORTED and the caller keeps the
// separate launches.
extern "C" int p3v_gemv_step(const p3v_gemv_args_t* a, const p3v_gemv_step_t* st, void* stream) {
  if (!a || !st || !a->W || !a->out) return P3V_ERR_ARG;
  const bool begin = st->tok != nullptr, end = st->next_tok != nullptr;
  if (begin == end) return P3V_ERR_ARG;                        // exactly one of the two ends
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return P3V_ERR_ARG;
  // one row only: that is where p3v_gemv itself runs this kernel (2 .. 8 rows go to k_gemv_mfma8, whose sums associate differently --
  // a folded step must stay bit-identical to the eager one)
  if (a->M != 1 || a->N % 2 || (a->K != 3072 && a->K != 8192) || p3v_tuning().gemv_variant != 3) return P3V_ERR_UNSUPPORTED;
  if (a->epilogue != P3V_EPI_NONE) return P3V_ERR_UNSUPPORTED;
  if (begin) {
    if (!st->embed_table || !st->x_out || !st->cos_t || !st->sin_t || !st->d_past || !st->cos_out || !st->sin_out || st->vocab <= 0) return P3V_ERR_ARG;
    if (((uintptr_t)st->embed_table | (uintptr_t)st->x_out) & 15) return P3V_ERR_ARG;
  } else {
    if (!a->x || !st->tok_out || !st->history || !st->d_step || !st->d_past || !st->ticket || !st->amax_ws) return P3V_ERR_ARG;
    if ((uintptr_t)st->amax_ws & 7) return P3V_ERR_ARG;
  }
  GemvP p = {a->x, a->W, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue, (a->N + 1) / 2};
  const GemvStepP sp = {st->tok, st->embed_table, st->vocab, st->x_out, st->cos_t, st->sin_t, st->d_past, st->cos_out, st->sin_out, st->tab_t,
                        st->half_dim, st->next_tok, st->tok_out, st->history, st->d_step, st->d_past, st->ticket, st->amax_ws, st->max_steps};
  hipStream_t s = (hipStream_t)stream;
  if (a->K == 3072) return begin ? launch_gemv3<1, 1, 6, STEP_BEGIN>(p, s, &sp) : launch_gemv3<1, 1, 6, STEP_END>(p, s, &sp);
  return begin ? launch_gemv3<1, 4, 4, STEP_BEGIN>(p, s, &sp) : launch_gemv3<1, 4, 4, STEP_END>(p, s, &sp);
}

extern "C" int p3v_gemv(const p3v_gemv_args_t* a, void* stream) {
  if (!a || !a->x || !a->W || !a->out) return P3V_ERR_ARG;
  if (a->M <= 0 || a->M > 16 || a->N <= 0 || a->K <= 0 || a->K % 8) return P3V_ERR_ARG;
  if (a->epilogue != P3V_EPI_NONE && a->epilogue != P3V_EPI_RESID_BF16 && a->epilogue != P3V_EPI_SILU_MUL &&
      a->epilogue != P3V_EPI_F32)
    return P3V_ERR_UNSUPPORTED;
  if (a->epilogue == P3V_EPI_RESID_BF16 && !a->resid) return P3V_ERR_ARG;
  const int mt = a->M <= 1 ? 1 : a->M <= 2 ? 2 : a->M <= 4 ? 4 : 8;
  GemvP p = {a->x, a->W, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue,
             a->epilogue == P3V_EPI_SILU_MUL ? a->N : (a->N + 1) / 2};
  hipStream_t s = (hipStream_t)stream;
  const int variant = p3v_tuning().gemv_variant;       // 1: generic kernel only; 3: streaming kernel where it applies
  if (variant == 3 && mt == 1 && a->N % 2 == 0 && (a->K == 3072 || a->K == 8192)) {
    if (a->K == 3072) return launch_gemv3<1, 1, 6>(p, s);
    return launch_gemv3<1, 4, 4>(p, s);                  // 8192 = 4 stages x 4 chunks: keeps 2 waves/SIMD resident
  }
  // 2 <= M <= 4: the same streaming kernel with MT activation rows in LDS (full-line weight loads, 4 v_dot2c per row and
  // 16-byte chunk): measured 2.19 vs 2.94 ms/step at B = 2 and 2.64 vs 3.26 at B = 4 against k_gemv_mfma; from M = 5 on the
  // matrix cores take over (at MT = 8 the VALU / LDS work per weight byte catches up: 4.72 ms/step at B = 8 against 4.08
  // with k_gemv_mfma and 3.27 with k_gemv_mfma8).
  const int rows_variant = p3v_tuning().gemv_rows;
  // smallest M the 8-row MFMA kernel takes: 2 since it became persistent with the RMSNorm scale in the epilogue (decode step
  // at ctx 2531, B = 2 / 3 / 4: 2.08 / 2.43 / 2.66 ms with the MT-row streaming kernel, 1.99 / 2.16 / 2.35 with this one);
  // P3V_GEMV8_MIN=5 restores the round-1 split
  const int rows8_min = p3v_tuning().gemv8_min;
  if (rows_variant && variant == 3 && a->M >= 2 && a->M <= 4 && a->M < rows8_min && a->N % 2 == 0 && a->epilogue != P3V_EPI_F32 &&
      (a->K == 3072 || a->K == 8192)) {
    if (a->K == 3072) return mt == 2 ? launch_gemv3<2, 1, 6>(p, s) : launch_gemv3<4, 1, 6>(p, s);
    return mt == 2 ? launch_gemv3<2, 4, 4>(p, s) : launch_gemv3<4, 4, 4>(p, s);
  }
  const int rows8 = p3v_tuning().gemv_mfma8;
  if (rows8 && a->M >= rows8_min && a->M <= 8) {
    const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
    switch (a->K) {                                      // K = NW waves x NST stages x 256
      case 1024: return silu ? launch_gemv_mfma8<true, 4, 1>(p, s) : launch_gemv_mfma8<false, 4, 1>(p, s);
      case 2048: return silu ? launch_gemv_mfma8<true, 4, 2>(p, s) : launch_gemv_mfma8<false, 4, 2>(p, s);
      case 3072: return silu ? launch_gemv_mfma8<true, 4, 3>(p, s) : launch_gemv_mfma8<false, 4, 3>(p, s);
      case 4096: return silu ? launch_gemv_mfma8<true, 8, 2>(p, s) : launch_gemv_mfma8<false, 8, 2>(p, s);
      case 8192: return silu ? launch_gemv_mfma8<true, 8, 4>(p, s) : launch_gemv_mfma8<false, 8, 4>(p, s);
      default: break;
    }
  }
  if (a->M >= 2 && a->K % (4 * 32 * GM_G) == 0 && !p3v_tuning().gemv_no_mfma) return launch_gemv_mfma(p, s);
  if (a->M > 8 || (size_t)mt * a->K * 2 > 160 * 1024 - 256) return P3V_ERR_UNSUPPORTED;
  switch (mt) {
    case 1: return launch_gemv<1>(p, s);
    case 2: return launch_gemv<2>(p, s);
    case 4: return launch_gemv<4>(p, s);
    default: return launch_gemv<8>(p, s);
  }
}
