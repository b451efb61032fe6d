// 4-bit group-64 affine weights (the reference's `quantize_model=True`: nn.quantize(model, group_size=64, bits=4),
// phi_3_vision_mlx.py:264,297-305 -> mx.quantized_matmul):  w[n, k] = scale[n, k/64] * q[n, k] + bias[n, k/64], q in 0..15.
// Decode-time projection y = x W^T streaming 0.5 byte per weight (+ 1/16 for scales and biases).
//
// Device layout (repacked once at load time from MLX's, see weights.q4_repack):
//   W4 [N, K/8] u32: the 8 nibbles of weights 8d .. 8d+7 sit at bits 0,16,4,20,8,24,12,28, so that
//        P_j = ((R >> 4j) & 0x000F000F) | 0x43004300      (one v_and_or_b32 after the shift)
//      is the bf16 PAIR (128 + q[2j], 128 + q[2j+1]) -- 0x4300 | q is exactly 128 + q in bf16 -- ready for v_dot2c_f32_bf16
//      against the activation pair (x[2j], x[2j+1]);
//   SB [N, K/64] u32: scale (bf16) | bias (bf16) << 16.
// Per 16-weight piece p of a row, with D = sum_k x_k (128 + q_k) and X = sum_k x_k (16 activations, shared by all rows,
// precomputed in LDS):   contribution = scale * (D - 128 X) + bias * X.
// Same streaming skeleton as k_gemv3: x (+RMSNorm) staged in LDS, (row pair, stage) pipeline with two register buffers,
// unconditional loads, branch-free body.  HBM-bound: N*K/2 + N*K/16 bytes per launch.
#include <stdlib.h>

#include <type_traits>

#include "p3v_common.h"
#include "p3v_dot_q4.h"
#include "p3v_gemv3_body.h"      // GemvStepP + the step-end helpers shared with the bf16 / e4m3 kernels

struct GemvQ4P {
  const bf16_t* x; const uint32_t* W; const uint32_t* sb; void* out; const bf16_t* resid; const bf16_t* norm_w;
  float eps;
  int M, N, K, epi, units;
};
typedef std::integral_constant<int, 0> QC0;
typedef std::integral_constant<int, 1> QC1;
// NST stages x NP 16-weight pieces per lane per row: K = NST * NP * 64 * 16
// STEP (p3v_gemv3_body.h): STEP_BEGIN / STEP_END carry the replayed greedy step's two ends, as gemv3_body does for bf16 weights
template <int NST, int NP, int STEP>
__device__ __forceinline__ void gemv3_q4_body(const GemvQ4P& p, int units_per_wave, int wpw, unsigned char* smem, float* red, const GemvStepP* sp) {
  constexpr int PIECES = NST * NP * 64;                 // 16-weight pieces per row
  constexpr int K = PIECES * 16, XCH = K / 8;           // 16-byte x chunks
  constexpr int XC = (XCH + 255) / 256;
  u32x4_t* xs = (u32x4_t*)smem;                         // [K] bf16 x (normalised)
  float* xsum = (float*)(smem + K * 2);                 // [PIECES] sum of the 16 activations of a piece
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool silu = p.epi == P3V_EPI_SILU_MUL, has_res = p.epi == P3V_EPI_RESID_BF16;
  const int u_begin = wave < wpw ? min(p.units, (blockIdx.x * wpw + wave) * units_per_wave) : p.units;   // (wpw: see p3v_gemv_wpw)
  const int u_end = min(p.units, u_begin + units_per_wave);
  const int n_st = (u_end - u_begin) * NST;

  u32x4_t xv[XC], gv[XC];
  const bf16_t* xrow[1] = {p.x};
  if (STEP == STEP_BEGIN) {
    int id = sp->tok[0];                                // (uniform: a scalar load)
    id = id < 0 ? 0 : (id >= sp->vocab ? sp->vocab - 1 : id);
    xrow[0] = sp->table + (size_t)id * K;
  }
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = min(tid + k * 256, XCH - 1);
    xv[k] = ((const u32x4_t*)xrow[0])[c];
    gv[k] = p.norm_w ? ((const u32x4_t*)p.norm_w)[c] : (u32x4_t){0, 0, 0, 0};
  }
  u32x2_t wbuf[2][2][NP];
  uint32_t sbuf[2][2][NP];
  uint32_t rbuf[2];
  auto issue = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int u = min(u_begin + gs / NST, p.units - 1), s = gs % NST;
    const int r0 = silu ? u : 2 * u, r1 = silu ? u + p.N : min(2 * u + 1, p.N - 1);
    const u32x2_t* w0 = (const u32x2_t*)(p.W + (size_t)r0 * (K / 8)) + s * NP * 64 + lane;
    const u32x2_t* w1 = (const u32x2_t*)(p.W + (size_t)r1 * (K / 8)) + s * NP * 64 + lane;
    const uint32_t* s0 = p.sb + (size_t)r0 * (K / 64) + ((s * NP * 64 + lane) >> 2);
    const uint32_t* s1 = p.sb + (size_t)r1 * (K / 64) + ((s * NP * 64 + lane) >> 2);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      wbuf[buf][0][j] = __builtin_nontemporal_load(w0 + j * 64);
      wbuf[buf][1][j] = __builtin_nontemporal_load(w1 + j * 64);
      sbuf[buf][0][j] = s0[j * 16];
      sbuf[buf][1][j] = s1[j * 16];
    }
    rbuf[buf] = has_res ? *(const uint32_t*)(p.resid + 2 * u) : 0u;
  };
  if (n_st > 0) issue(0, QC0{});

  float r = 1.f;
  if (p.norm_w) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < XC; ++k)
      if (tid + k * 256 < XCH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf16lo(xv[k][j]), b = bf16hi(xv[k][j]); ss += a * a + b * b; }
      }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    r = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + p.eps);
  }
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = tid + k * 256;
    if (c < XCH) {
      u32x4_t o = xv[k];
      if (p.norm_w) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = rms_pair(xv[k][j], r, gv[k][j]);
      }
      xs[c] = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int pc = tid; pc < PIECES; pc += 256) {           // X of every piece, from the (rounded) activations the dots see
    const u32x4_t a = xs[2 * pc], b = xs[2 * pc + 1];
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += (bf16lo(a[j]) + bf16hi(a[j])) + (bf16lo(b[j]) + bf16hi(b[j]));
    xsum[pc] = t;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float a0 = 0.f, a1 = 0.f;
  ArgMaxVI best[1] = {ArgMaxVI{-INFINITY, 0x7fffffff}};      // STEP_END: this wave's arg-max candidate (lane 0's copy counts)
  auto compute = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int s = gs % NST;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int pc = (s * NP + j) * 64 + lane;
      const u32x4_t xa = xs[2 * pc], xb = xs[2 * pc + 1];
      const float X = xsum[pc], X128 = 128.f * X;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u32x2_t w = wbuf[buf][h][j];
        const float D = dot8_q4(w[1], xb, dot8_q4(w[0], xa, 0.f));
        const uint32_t sb = sbuf[buf][h][j];
        const float c = bf16lo(sb) * (D - X128) + bf16hi(sb) * X;
        if (h == 0) a0 += c; else a1 += c;
      }
    }
    if (s == NST - 1) {
      const int u = u_begin + gs / NST;
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      if (lane == 0) {
        if (silu) {
          const float g = bf16_round(a0), up = bf16_round(a1);
          ((bf16_t*)p.out)[u] = f32_to_bf16(bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * up);
        } else if (p.epi == P3V_EPI_F32) {
          ((float*)p.out)[2 * u] = a0;
          ((float*)p.out)[2 * u + 1] = a1;
        } else {
          float v0 = a0, v1 = a1;
          if (has_res) { v0 = bf16lo(rbuf[buf]) + bf16_round(v0); v1 = bf16hi(rbuf[buf]) + bf16_round(v1); }
          *(uint32_t*)((bf16_t*)p.out + 2 * u) = pack_bf16x2(v0, v1);
          if (STEP == STEP_END) {                              // on the values just stored (bf16)
            amax_take(best[0], bf16_round(v0), 2 * u);
            if (2 * u + 1 < p.N) amax_take(best[0], bf16_round(v1), 2 * u + 1);
          }
        }
      }
      a0 = a1 = 0.f;
    }
  };
  int gs = 0;
  while (gs + 2 < n_st) {
    issue(gs + 1, QC1{}); compute(gs, QC0{});
    issue(gs + 2, QC0{}); compute(gs + 1, QC1{});
    gs += 2;
  }
  if (gs + 1 < n_st) {
    issue(gs + 1, QC1{}); compute(gs, QC0{}); compute(gs + 1, QC1{});
  } else if (gs < n_st) {
    compute(gs, QC0{});
  }
  if (STEP == STEP_BEGIN && blockIdx.x == 0) gemv_step_begin_tail<1>(sp, xrow, 1, XCH, tid);
  if (STEP == STEP_END) gemv_step_end_tail<1>(sp, best, 1, (int)blockIdx.x, tid);
}

template <int NST, int NP>
__global__ void __launch_bounds__(256) k_gemv3_q4(GemvQ4P p, int units_per_wave, int wpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4];
  gemv3_q4_body<NST, NP, STEP_NONE>(p, units_per_wave, wpw, smem, red, nullptr);
}

template <int NST, int NP, int STEP>
__global__ void __launch_bounds__(256) k_gemv3_q4_step(GemvQ4P p, int units_per_wave, int wpw, GemvStepP sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4];
  gemv3_q4_body<NST, NP, STEP>(p, units_per_wave, wpw, smem, red, &sp);
}

// W4 / SB -> bf16 [rows, K] (prefill and batched decode run the bf16 kernels on a dequantised scratch copy)
__global__ void __launch_bounds__(256) k_dequant_q4(const uint32_t* __restrict__ w, const uint32_t* __restrict__ sb,
                                                    u32x4_t* __restrict__ out, int kd, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // dword index: row * kd + d
  if (i >= total) return;
  const long row = i / kd;
  const int d = (int)(i - row * kd);
  const uint32_t r = w[i], s = sb[row * (kd / 8) + (d >> 3)];
  const float sc = bf16lo(s), bi = bf16hi(s);
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float q0 = (float)((r >> (4 * j)) & 15u), q1 = (float)((r >> (4 * j + 16)) & 15u);
    o[j] = pack_bf16x2(sc * q0 + bi, sc * q1 + bi);
  }
  out[i] = o;
}

extern "C" int p3v_dequant_q4(const uint32_t* w4, const uint32_t* sb, uint16_t* out_bf16, int rows, int K, void* stream) {
  if (!w4 || !sb || !out_bf16 || rows <= 0 || K <= 0 || K % 64) return P3V_ERR_ARG;
  const long total = (long)rows * (K / 8);
  hipLaunchKernelGGL(k_dequant_q4, dim3(p3v_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w4, sb, (u32x4_t*)out_bf16, K / 8, total);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <int NST, int NP, int STEP = STEP_NONE>
static int launch_gemv3_q4(const GemvQ4P& p, hipStream_t s, const GemvStepP* sp = nullptr) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return P3V_ERR_HIP;
    n_cu = pr.multiProcessorCount;
  }
  const int wpc = p3v_tuning().gemv_q4_wpc;   // waves per CU
  int upw = p3v_cdiv(p.units, n_cu * wpc);
  if (upw < 1) upw = 1;
  const int waves = p3v_cdiv(p.units, upw);
  const int wpw = p3v_gemv_wpw(waves, n_cu, p3v_tuning().gemv_wpw);
  if constexpr (STEP != STEP_NONE) {
    if (p3v_cdiv(waves, wpw) > P3V_GEMV_STEP_MAX_WG) return P3V_ERR_UNSUPPORTED;        // (amax_ws holds one candidate per workgroup)
    hipLaunchKernelGGL((k_gemv3_q4_step<NST, NP, STEP>), dim3(p3v_cdiv(waves, wpw)), dim3(256), (size_t)p.K * 2 + (size_t)p.K / 4, s, p, upw, wpw, *sp);
  } else {
    hipLaunchKernelGGL((k_gemv3_q4<NST, NP>), dim3(p3v_cdiv(waves, wpw)), dim3(256), (size_t)p.K * 2 + (size_t)p.K / 4, s, p, upw, wpw);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// p3v_gemv_step on MLX 4-bit weights: the first / last projection of a replayed greedy step with p3v_step_begin / p3v_step_end folded in
// (one row, K = 3072 or 8192, no epilogue); anything else reports P3V_ERR_UNSUPPORTED and the caller keeps the separate launches.
extern "C" int p3v_gemv_q4_step(const p3v_gemv_q4_args_t* a, const p3v_gemv_step_t* st, void* stream) {
  if (!a || !st || !a->W || !a->sb || !a->out) return P3V_ERR_ARG;
  const bool begin = st->tok != nullptr, end = st->next_tok != nullptr;
  if (begin == end) return P3V_ERR_ARG;                        // exactly one of the two ends
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return P3V_ERR_ARG;
  if (a->M != 1 || a->N % 2 || (a->K != 3072 && a->K != 8192) || a->epilogue != P3V_EPI_NONE) return P3V_ERR_UNSUPPORTED;
  if (begin) {
    if (!st->embed_table || !st->x_out || !st->cos_t || !st->sin_t || !st->d_past || !st->cos_out || !st->sin_out || st->vocab <= 0) return P3V_ERR_ARG;
    if (((uintptr_t)st->embed_table | (uintptr_t)st->x_out) & 15) return P3V_ERR_ARG;
  } else {
    if (!a->x || !st->tok_out || !st->history || !st->d_step || !st->d_past || !st->ticket || !st->amax_ws) return P3V_ERR_ARG;
    if ((uintptr_t)st->amax_ws & 7) return P3V_ERR_ARG;
  }
  const GemvQ4P p = {a->x, a->W, a->sb, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue, a->N / 2};
  const GemvStepP sp = {st->tok, st->embed_table, st->vocab, st->x_out, st->cos_t, st->sin_t, st->d_past, st->cos_out, st->sin_out, st->tab_t,
                        st->half_dim, st->next_tok, st->tok_out, st->history, st->d_step, st->d_past, st->ticket, st->amax_ws, st->max_steps};
  hipStream_t s = (hipStream_t)stream;
  if (a->K == 3072) return begin ? launch_gemv3_q4<1, 3, STEP_BEGIN>(p, s, &sp) : launch_gemv3_q4<1, 3, STEP_END>(p, s, &sp);
  return begin ? launch_gemv3_q4<2, 4, STEP_BEGIN>(p, s, &sp) : launch_gemv3_q4<2, 4, STEP_END>(p, s, &sp);
}

// ---------------------------------------------------------------------------------------------------------------------
// 2 .. 16 rows of x on the 4-bit weights (round 6): the reference runs QuantizedLinear at every batch size
// (phi_3_vision_mlx.py:296; README batched 4-bit benchmark), rounds 3-5 dequantised the WHOLE matrix into a bf16 scratch per call
// (2 launches, 5 x the 4-bit bytes moved).  This is k_gemm_rows (p3v_gemm_rows.hip) on the packed weights: the lane that held a
// 16-byte bf16 chunk of a weight line now loads 16 bytes = 32 WEIGHTS of the row (eight lanes cover the 128-byte line that holds a
// row's 256-weight block: whole lines per instruction, 0.5 byte per weight from HBM), all of one 64-weight group (dword j of the
// lane = weights 64 g + 32 h + 8 j ..+7 of the block: group g), so ONE scale | bias word per (lane, block, row set) dequantises them in
// registers: (1024 + q) as an fp16 pair straight from the nibbles (0x6400 | q), - 1024, * scale + bias with v_pk_fma_f16 -- two
// weights per instruction, 11 significant bits (the bf16 scratch of the old path kept 8) -- and the products run on the fp16 MFMA
// against x converted once to fp16 fragments in the matching k-order (column m, k-slot g of MFMA j: x[m][64 g + 32 h' + 8 j ..+7]).
// K = 3072: 4 waves x 3 blocks; K = 8192: 8 waves x 4 blocks.  Epilogues and the cross-wave exchange as k_gemm_rows.
typedef _Float16 q4h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 q4h8_t __attribute__((ext_vector_type(8)));
struct RowsQ4P {
  const bf16_t* x; const uint32_t* W; const uint32_t* sb; void* out; const bf16_t* resid;
  int M, N, K, epi, n_sets;
};
typedef std::integral_constant<int, 2> QC2;
typedef std::integral_constant<int, 3> QC3;

__device__ __forceinline__ q4h8_t q4_dequant8(uint32_t d, q4h2_t s2, q4h2_t b2) {
  const q4h2_t k1024 = {(_Float16)1024.f, (_Float16)1024.f};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t pw = ((d >> (4 * i)) & 0x000F000Fu) | 0x64006400u;       // fp16 pair (1024 + q[2i], 1024 + q[2i + 1])
    asm("" : "+v"(pw));                                               // (hipcc folds bit_casts of freshly built words: see dot8)
    const q4h2_t q = __builtin_bit_cast(q4h2_t, pw) - k1024;
    o[i] = __builtin_bit_cast(uint32_t, q * s2 + b2);
  }
  const u32x4_t v = {o[0], o[1], o[2], o[3]};
  return __builtin_bit_cast(q4h8_t, v);
}

template <bool SILU, int NW, int NST>
__global__ void __launch_bounds__(NW * 64) k_gemm_rows_q4(RowsQ4P p) {
  constexpr int KQ = NST * 256;                                  // a wave's K quarter (NST 256-weight blocks)
  __shared__ float cpart[2][NW * 2 * 8 * 16];                    // [parity][wave][row set][weight row][x row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int r8 = li >> 1, hbit = lane & 1;
  const int k_lo = wave * KQ;
  const int kd = p.K / 8, kg = p.K / 64;                         // dwords / groups per weight row
  auto row_ptrs = [&](int set, const uint32_t*& w0, const uint32_t*& w1, const uint32_t*& s0, const uint32_t*& s1) {
    const int n_base = set * (SILU ? 8 : 16);
    const int row0 = min(n_base + r8, p.N - 1);
    const int row1 = SILU ? p.N + row0 : min(n_base + 8 + r8, p.N - 1);
    w0 = p.W + (size_t)row0 * kd + k_lo / 8 + 4 * (2 * g + hbit);
    w1 = p.W + (size_t)row1 * kd + k_lo / 8 + 4 * (2 * g + hbit);
    s0 = p.sb + (size_t)row0 * kg + k_lo / 64 + g;
    s1 = p.sb + (size_t)row1 * kg + k_lo / 64 + g;
  };
  // One set of weight registers, each stage refilled with the same stage of the NEXT set as it is consumed.  (A ring of 2 - 3 sets was
  // built and measured slower; the register budget is x's: 96 VGPRs of fragments in both k-orders.)
  constexpr int D = 1;
  const int stride = (int)gridDim.x;
  int set = blockIdx.x;
  u32x4_t wa[D][NST][2];
  uint32_t sv[D][NST][2];
  auto issue = [&](int which, auto slotc, auto stc) {
    constexpr int slot = decltype(slotc)::value, st = decltype(stc)::value;
    const uint32_t *a0, *a1, *b0, *b1;
    row_ptrs(which, a0, a1, b0, b1);
    wa[slot][st][0] = __builtin_nontemporal_load((const u32x4_t*)(a0 + st * 32));
    wa[slot][st][1] = __builtin_nontemporal_load((const u32x4_t*)(a1 + st * 32));
    sv[slot][st][0] = b0[st * 4];
    sv[slot][st][1] = b1[st * 4];
  };
  auto issue_set = [&](int which, auto slotc) {
    issue(which, slotc, QC0{});
    if constexpr (NST > 1) issue(which, slotc, QC1{});
    if constexpr (NST > 2) issue(which, slotc, QC2{});
    if constexpr (NST > 3) issue(which, slotc, QC3{});
  };
  static_assert(NST <= 4, "unrolled for at most four blocks");
  issue_set(set, QC0{});

  // (The input RMSNorm is NOT fused here: built and measured -- every one of the 192-256 workgroups normalising its own copy of the 2 .. 16
  //  rows costs more than the one p3v_rmsnorm launch it saves: B = 8 at 512 keys 2.39 ms per step fused against 2.13 with the launch.)
  q4h8_t xf[NST][4][2];
  {
    const bf16_t* xr = p.x + (size_t)min(li, p.M - 1) * p.K + k_lo + 64 * g;
#pragma unroll
    for (int blk = 0; blk < NST; ++blk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4_t v = *(const u32x4_t*)(xr + blk * 256 + 32 * h + 8 * j);
          q4h8_t f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[2 * e] = (_Float16)bf16lo(v[e]); f[2 * e + 1] = (_Float16)bf16hi(v[e]); }
          xf[blk][j][h] = f;
        }
  }

  int par = 0;
  // (two straight-line copies: with a further set to request, and the last one -- see k_gemv_mfma8, p3v_gemv.hip)
  auto do_set = [&](auto refillc) {
    constexpr int slot = 0;
    constexpr bool REFILL = decltype(refillc)::value;
    const auto slotc = QC0{};
    const int refill = set + stride;
    constexpr int ITEMS = (SILU ? 8 : 16) * 16;                  // (output column, x row) pairs of a set: at most one per thread
    static_assert(ITEMS <= NW * 64, "one epilogue item per thread");
    const int e_R = tid & 7, e_sub = SILU ? 0 : (tid >> 3) & 1, e_m = SILU ? tid >> 3 : tid >> 4;
    const int e_n = set * (SILU ? 8 : 16) + e_sub * 8 + e_R;
    const bool e_live = tid < ITEMS && e_m < p.M && e_n < p.N;
    const size_t e_o = (size_t)e_m * p.N + e_n;
    const bool e_has = !SILU && p.epi == P3V_EPI_RESID_BF16;      // the residual element goes out ahead of the refills (see k_gemv8_q4)
    uint32_t e_res = (e_has ? p.resid : p.x)[e_has && e_live ? e_o : 0];
    f32x4_t acc[2][2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) acc[s_][0] = acc[s_][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st < NST) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const _Float16 sc = (_Float16)bf16lo(sv[slot][st][s_]), bi = (_Float16)bf16hi(sv[slot][st][s_]);
          const q4h2_t s2 = {sc, sc}, b2 = {bi, bi};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const q4h8_t a = q4_dequant8(wa[slot][st][s_][j], s2, b2);
            acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[st][j][0], acc[s_][0], 0, 0, 0);
            acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[st][j][1], acc[s_][1], 0, 0, 0);
          }
        }
        if constexpr (REFILL) { issue(refill, slotc, stc); __builtin_amdgcn_sched_barrier(0); }   // (pinned: the scheduler otherwise sinks the refills to the end of the set)
      }
    };
    step(QC0{}); step(QC1{}); step(QC2{}); step(QC3{});
    float* cp = cpart[par];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      cp[((wave * 2 + s_) * 8 + 2 * g) * 16 + li] = acc[s_][0][0] + acc[s_][1][1];
      cp[((wave * 2 + s_) * 8 + 2 * g + 1) * 16 + li] = acc[s_][0][2] + acc[s_][1][3];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (e_live) {
      const int R = e_R, sub = e_sub, m = e_m;
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        v0 += cp[((w * 2 + sub) * 8 + R) * 16 + m];
        if (SILU) v1 += cp[((w * 2 + 1) * 8 + R) * 16 + m];
      }
      const size_t o = e_o;
      if (SILU) {
        const float gt = bf16_round(v0), up = bf16_round(v1);
        ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
      } else if (p.epi == P3V_EPI_F32) {
        ((float*)p.out)[o] = v0;
      } else if (p.epi == P3V_EPI_RESID_BF16) {
        asm volatile("" : "+v"(e_res));
        ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32((bf16_t)e_res) + bf16_round(v0));
      } else {
        ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
      }
    }
    set += stride;
    par ^= 1;
  };
  while (set + stride < p.n_sets) do_set(std::true_type{});
  do_set(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// 2 .. 8 rows of x on the 4-bit weights: k_gemv_mfma8 (p3v_gemv.hip) with the packed weights dequantised in front of the fp16 MFMA.
// What k_gemm_rows_q4 above pays per workgroup -- every lane fetching its own 16-byte pieces of 16 x rows (64 scattered requests per
// load instruction, ~2.5 us of L1 request time per workgroup) and both k-orders of x in registers (96 VGPRs: no room for a deeper weight
// ring) -- this one does not: the wave's K quarter of x is read in whole 1 KB runs, normalised (the input RMSNorm rides along: sums of
// squares per wave -> LDS -> one barrier -> bf16(bf16(x r) g), the value p3v_rmsnorm writes, exact in fp16), parked in LDS as fp16 in
// fragment order, and each MFMA reads its B operand from there with one ds_read_b128; with at most 8 rows the 16 MFMA columns are
// (x row m, k-half h') pairs, so ONE product per weight fragment carries both k-orders (column 2 m + h', as k_gemv_mfma8).
template <bool SILU, int NW, int NST>
__global__ void __launch_bounds__(NW * 64) k_gemv8_q4(RowsQ4P p, const bf16_t* norm_w, float eps) {
  constexpr int KQ = NST * 256, K = KQ * NW, XS = K * 2 + 32, NCH = KQ / 8;     // wave's K quarter, LDS row stride (bytes: 2 x 16 B mod 256), its 8-element chunks
  constexpr int ROWS = SILU ? 8 : 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[NW][8];
  __shared__ float cpart[2][NW * 2 * 8 * 8];                     // [parity][wave][row set][weight row][x row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int r8 = (lane & 15) >> 1, hbit = lane & 1;
  const int k_lo = wave * KQ;
  const int kd = K / 8, kg = K / 64;
  auto row_ptrs = [&](int set, const uint32_t*& w0, const uint32_t*& w1, const uint32_t*& s0, const uint32_t*& s1) {
    const int n_base = set * ROWS;
    const int row0 = min(n_base + r8, p.N - 1);
    const int row1 = SILU ? p.N + row0 : min(n_base + 8 + r8, p.N - 1);
    w0 = p.W + (size_t)row0 * kd + k_lo / 8 + 4 * (2 * g + hbit);
    w1 = p.W + (size_t)row1 * kd + k_lo / 8 + 4 * (2 * g + hbit);
    s0 = p.sb + (size_t)row0 * kg + k_lo / 64 + g;
    s1 = p.sb + (size_t)row1 * kg + k_lo / 64 + g;
  };
  const int stride = (int)gridDim.x;
  int set = blockIdx.x;
  constexpr int D = 1;                                           // (rings of 2 .. 4 sets were built and measured: no faster)
  u32x4_t wa[D][NST][2];
  uint32_t sv[D][NST][2];
  auto issue = [&](int which, auto slotc, auto stc) {
    constexpr int slot = decltype(slotc)::value, st = decltype(stc)::value;
    const uint32_t *a0, *a1, *b0, *b1;
    row_ptrs(which, a0, a1, b0, b1);
    wa[slot][st][0] = __builtin_nontemporal_load((const u32x4_t*)(a0 + st * 32));
    wa[slot][st][1] = __builtin_nontemporal_load((const u32x4_t*)(a1 + st * 32));
    sv[slot][st][0] = b0[st * 4];
    sv[slot][st][1] = b1[st * 4];
  };
  auto issue_set = [&](int which, auto slotc) {
    issue(which, slotc, QC0{});
    if constexpr (NST > 1) issue(which, slotc, QC1{});
    if constexpr (NST > 2) issue(which, slotc, QC2{});
    if constexpr (NST > 3) issue(which, slotc, QC3{});
  };
  static_assert(NST <= 4, "unrolled for at most four blocks");

  // ---- x quarter (8 rows x NCH chunks, lanes over chunks: 1 KB runs), then the ring's weights
  unsigned char* xslice = smem + k_lo * 2;
  {
    u32x4_t xv[8][2], gv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = min(lane + 64 * k, NCH - 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) xv[m][k] = *(const u32x4_t*)(p.x + (size_t)min(m, p.M - 1) * K + k_lo + 8 * c);
      gv[k] = norm_w ? *(const u32x4_t*)(norm_w + k_lo + 8 * c) : (u32x4_t){0, 0, 0, 0};
    }
    issue_set(set, QC0{});
    float r[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = 1.f;
    if (norm_w) {                                                // workgroup-uniform
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (lane + 64 * k < NCH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float a = bf16lo(xv[m][k][j]), b = bf16hi(xv[m][k][j]); ss += a * a + b * b; }
          }
        ss = wave_sum(ss);
        if (lane == 0) red[wave][m] = ss;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][m];
        r[m] = rsqrtf(t / (float)K + eps);
      }
    }
    // fragment order inside a 256-element block: the 16-byte piece of elements 64 g + 32 h + 8 j ..+7 sits at slot 8 j + 2 g + h, so
    // that the 16 lanes of an MFMA row group (x row r8 = 0..7, h) read 16 different 16-byte slots (row stride = 2 slots mod 16)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = lane + 64 * k;
      if (c < NCH) {
        const int blk = c >> 5, w = c & 31, slot = 8 * (w & 3) + 2 * (w >> 3) + ((w >> 2) & 1);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t hw = norm_w ? rms_pair(xv[m][k][j], r[m], gv[k][j]) : xv[m][k][j];
            const q4h2_t f = {(_Float16)bf16lo(hw), (_Float16)bf16hi(hw)};
            o[j] = m < p.M ? __builtin_bit_cast(uint32_t, f) : 0u;
          }
          *(u32x4_t*)(xslice + m * XS + blk * 512 + slot * 16) = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the slice is wave-private: no barrier

  const unsigned char* xrow = xslice + r8 * XS + (2 * g + hbit) * 16;
  const bool odd = lane & 1;
  int par = 0;
  // Two straight-line copies of the set body (as k_gemv_mfma8, p3v_gemv.hip): with a further set to request stage by stage, and the
  // last one.  A branch around the refills makes the compiler's counted vmcnt ignore them -- every stage then waits for loads issued a
  // moment ago (measured: 2 us per set instead of 1.5).
  auto do_set = [&](auto refillc) {
    constexpr int slot = 0;
    constexpr bool REFILL = decltype(refillc)::value;
    const auto slotc = QC0{};
    const int refill = set + stride;
    // the residual element of this thread's output goes out BEFORE this set's refills: read in the epilogue it would be the youngest
    // load in flight and its wait (vmcnt(0)) would drain the whole ring at every set
    const int e_R = tid & 7, e_m = (tid >> 3) & 7, e_sub = tid >> 6;
    const int e_n = set * ROWS + e_sub * 8 + e_R;
    const bool e_live = tid < (SILU ? 64 : 128) && e_m < p.M && e_n < p.N;
    const size_t e_o = (size_t)e_m * p.N + e_n;
    const bool e_has = !SILU && p.epi == P3V_EPI_RESID_BF16;    // (an unconditional load: element 0 of x where there is no residual)
    uint32_t e_res = (e_has ? p.resid : p.x)[e_has && e_live ? e_o : 0];
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st < NST) {
        const _Float16 sc0 = (_Float16)bf16lo(sv[slot][st][0]), bi0 = (_Float16)bf16hi(sv[slot][st][0]);
        const _Float16 sc1 = (_Float16)bf16lo(sv[slot][st][1]), bi1 = (_Float16)bf16hi(sv[slot][st][1]);
        const q4h2_t s20 = {sc0, sc0}, b20 = {bi0, bi0}, s21 = {sc1, sc1}, b21 = {bi1, bi1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const q4h8_t xb = *(const q4h8_t*)(xrow + st * 512 + j * 128);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q4_dequant8(wa[slot][st][0][j], s20, b20), xb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q4_dequant8(wa[slot][st][1][j], s21, b21), xb, acc1, 0, 0, 0);
        }
        if constexpr (REFILL) { issue(refill, slotc, stc); __builtin_amdgcn_sched_barrier(0); }   // (pinned: the scheduler otherwise sinks the refills to the end of the set)
      }
    };
    step(QC0{}); step(QC1{}); step(QC2{}); step(QC3{});
    // C[4 (lane >> 4) + e][column 2 m + h']: e = h' and h' + 2 are weight rows 2 (lane >> 4), + 1 over k-half h'; the halves meet by DPP
    float e[2][2] = {{odd ? acc0[1] : acc0[0], odd ? acc0[3] : acc0[2]}, {odd ? acc1[1] : acc1[0], odd ? acc1[3] : acc1[2]}};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) e[t][u] += P3V_DPP_F32(e[t][u], 0xB1);        // quad_perm [1,0,3,2]
    float* cp = cpart[par];
    if (!odd) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) cp[((wave * 2 + t) * 8 + 2 * g + u) * 8 + r8] = e[t][u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // (the weight loads of the sets ahead stay in flight across it)
    if (e_live) {
      const int R = e_R, m = e_m, sub = e_sub;
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        v0 += cp[((w * 2 + sub) * 8 + R) * 8 + m];
        if (SILU) v1 += cp[((w * 2 + 1) * 8 + R) * 8 + m];
      }
      const size_t o = e_o;
      if (SILU) {
        const float gt = bf16_round(v0), up = bf16_round(v1);
        ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
      } else if (p.epi == P3V_EPI_F32) {
        ((float*)p.out)[o] = v0;
      } else if (p.epi == P3V_EPI_RESID_BF16) {
        asm volatile("" : "+v"(e_res));                          // (the value is first looked at HERE: no wait for it at the top of the set)
        ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32((bf16_t)e_res) + bf16_round(v0));
      } else {
        ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
      }
    }
    set += stride;
    par ^= 1;
  };
  while (set + stride < p.n_sets) do_set(std::true_type{});
  do_set(std::false_type{});
}

template <bool SILU, int NW, int NST>
static int launch_gemv8_q4(const RowsQ4P& p, const bf16_t* norm_w, float eps, hipStream_t s) {
  const size_t lds = (size_t)8 * (p.K * 2 + 32);
  static bool attr_set = false;
  if (!attr_set && lds > 40 * 1024) {
    if (hipFuncSetAttribute((const void*)k_gemv8_q4<SILU, NW, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8704) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const int per = p3v_cdiv(p.n_sets, p3v_tuning().gemv_q4_rows_wgs), gx = p3v_cdiv(p.n_sets, per);
  hipLaunchKernelGGL((k_gemv8_q4<SILU, NW, NST>), dim3(gx), dim3(NW * 64), lds, s, p, norm_w, eps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <bool SILU, int NW, int NST>
static int launch_rows_q4(const RowsQ4P& p, hipStream_t s) {
  const int per = p3v_cdiv(p.n_sets, p3v_tuning().gemv_q4_rows_wgs), gx = p3v_cdiv(p.n_sets, per);
  hipLaunchKernelGGL((k_gemm_rows_q4<SILU, NW, NST>), dim3(gx), dim3(NW * 64), 0, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

extern "C" int p3v_gemv_q4(const p3v_gemv_q4_args_t* a, void* stream) {
  if (!a || !a->x || !a->W || !a->sb || !a->out) return P3V_ERR_ARG;
  if (a->M >= 2 && a->M <= 16 && (a->K == 3072 || a->K == 8192) && (!a->norm_w || a->M <= 8) && a->N > 0) {
    // 2 .. 8 rows: k_gemv8_q4 (input RMSNorm optional); 9 .. 16: k_gemm_rows_q4 (the caller normalises)
    const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
    if (a->N % (silu ? 8 : 16)) return P3V_ERR_UNSUPPORTED;
    if (a->epilogue != P3V_EPI_NONE && a->epilogue != P3V_EPI_RESID_BF16 && !silu && a->epilogue != P3V_EPI_F32) return P3V_ERR_UNSUPPORTED;
    if (a->epilogue == P3V_EPI_RESID_BF16 && !a->resid) return P3V_ERR_ARG;
    if (((uintptr_t)a->x | (uintptr_t)a->W) & 15) return P3V_ERR_ARG;
    const RowsQ4P q = {a->x, a->W, a->sb, a->out, a->resid, a->M, a->N, a->K, a->epilogue, a->N / (silu ? 8 : 16)};
    hipStream_t s = (hipStream_t)stream;
    if (a->M <= 8 && p3v_tuning().gemv_q4_rows8) {
      if (a->norm_w && ((uintptr_t)a->norm_w & 15)) return P3V_ERR_ARG;
if (a->K == 3072) return silu ? launch_gemv8_q4<true, 4, 3>(q, a->norm_w, a->norm_eps, s) : launch_gemv8_q4<false, 4, 3>(q, a->norm_w, a->norm_eps, s);
      return silu ? launch_gemv8_q4<true, 8, 4>(q, a->norm_w, a->norm_eps, s) : launch_gemv8_q4<false, 8, 4>(q, a->norm_w, a->norm_eps, s);
    }
    if (a->norm_w) return P3V_ERR_UNSUPPORTED;
    if (a->K == 3072) return silu ? launch_rows_q4<true, 4, 3>(q, s) : launch_rows_q4<false, 4, 3>(q, s);
    return silu ? launch_rows_q4<true, 8, 4>(q, s) : launch_rows_q4<false, 8, 4>(q, s);
  }
  if (a->M != 1 || a->N <= 0 || a->N % 2 || (a->K != 3072 && a->K != 8192)) return P3V_ERR_UNSUPPORTED;
  if (a->epilogue != P3V_EPI_NONE && a->epilogue != P3V_EPI_RESID_BF16 && a->epilogue != P3V_EPI_SILU_MUL &&
      a->epilogue != P3V_EPI_F32)
    return P3V_ERR_UNSUPPORTED;
  if (a->epilogue == P3V_EPI_RESID_BF16 && !a->resid) return P3V_ERR_ARG;
  GemvQ4P p = {a->x, a->W, a->sb, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue,
               a->epilogue == P3V_EPI_SILU_MUL ? a->N : a->N / 2};
  hipStream_t s = (hipStream_t)stream;
  return a->K == 3072 ? launch_gemv3_q4<1, 3>(p, s) : launch_gemv3_q4<2, 4>(p, s);
}
