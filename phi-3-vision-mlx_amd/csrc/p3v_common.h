// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere; bf16 is carried as raw uint16_t in memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p3v.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte load unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define P3V_WAVE 64
#define P3V_MAX_DEVICES 64          // per-device caches of launch properties

#define P3V_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return P3V_ERR_LAUNCH;             \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }

// round-to-nearest-even through the native conversion (v_cvt_pk_bf16_f32 on gfx950: one VALU op per PAIR,
// where the integer formulation costs ~10 -- it was the largest VALU item of the attention softmax)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_t r = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// _rotate_half on one (d, d + hd/2) pair of the bf16 projection outputs a, b (phi.py:419-423, 443-452: q * cos + rotate_half(q) * sin
// in fp32), the query scale applied before the one rounding that follows.  ONE definition for the stand-alone kernel
// (p3v_rope_kv_append) and the fused qkv epilogue (p3v_gemm_qkv.h), with the products pinned as separately rounded values (the empty
// asm keeps hipcc's fp-contract from fusing one of them into the subtraction -- it chose differently in the two kernels: 1e-5 of
// the elements apart by one ulp): multiply, multiply, add, as the reference's array expression evaluates it.
__device__ __forceinline__ void p3v_rope_pair(float a, float b, float cs, float sn, float qs, float& o1, float& o2) {
  float ac = a * cs, bs = b * sn, bc = b * cs, as = a * sn;
  asm volatile("" : "+v"(ac), "+v"(bs), "+v"(bc), "+v"(as));
  o1 = (ac - bs) * qs;
  o2 = (bc + as) * qs;
}

// sigmoid of the quick-GELU epilogue (the ViT's fc1, fp32 tower): v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division (~10 VALU
// instructions per element: that epilogue spent 30 us per launch on it, tools/scratch/epi_abl.py).  The decoder's SiLU keeps the
// exact division: its result is rounded to bf16 at once, where the oracle's torch.sigmoid and a 1-ulp reciprocal part ways in
// ~2^-15 of the elements -- enough to move the heavy-tailed fixtures (outlier channels of gain 64) past their measured bound, and
// the gate_up epilogue gains nothing from it (205 vs 203 us).
__device__ __forceinline__ float p3v_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// nn.RMSNorm = mx.fast.rms_norm (phi.py:478-479, 571): w * astype(x * rsqrt(mean x^2 + eps), bf16) -- the normalised value is
// rounded to bf16 BEFORE the weight multiply, which rounds again (HF's Phi3RMSNorm does the same).  One bf16 pair at a time.
__device__ __forceinline__ uint32_t rms_pair(uint32_t x2, float r, uint32_t g2) {
  const uint32_t n = pack_bf16x2(bf16lo(x2) * r, bf16hi(x2) * r);
  return pack_bf16x2(bf16lo(n) * bf16lo(g2), bf16hi(n) * bf16hi(g2));
}

// cross-row all-reduce over the four 16-lane rows of a wave (lanes sharing lane & 15), on gfx950's
// v_permlane{16,32}_swap: swap(x, x) leaves {row r, row r^1} pairs in the two results, so one max / add finishes a
// butterfly step without the LDS crossbar latency of ds_bpermute.
// (The two integer results pass through an empty asm before they are reinterpreted as floats: hipcc 7.2 otherwise
// folds bitcast(result 1) into bitcast(result 0).)
__device__ __forceinline__ void rows_swap32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  unsigned x = r[0], y = r[1];
  asm("" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void rows_swap16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  unsigned x = r[0], y = r[1];
  asm("" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float rows_max(float v) {
  float a, b;
  rows_swap32(v, a, b); v = fmaxf(a, b);
  rows_swap16(v, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float rows_sum(float v) {
  float a, b;
  rows_swap32(v, a, b); v = a + b;
  rows_swap16(v, a, b); return a + b;
}

// full-wave all-reduce without the LDS crossbar: xor-1 / xor-2 inside a quad and the two rotations inside a row of 16
// lanes are DPP modifiers of the add / max itself, the last two steps are the row swaps above.  (ds_bpermute, which
// __shfl_xor compiles to, is a ~100-cycle LDS round trip per step, six dependent ones per reduction.)
#define P3V_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
  v += P3V_DPP_F32(v, 0xB1);      // quad_perm [1,0,3,2]
  v += P3V_DPP_F32(v, 0x4E);      // quad_perm [2,3,0,1]
  v += P3V_DPP_F32(v, 0x124);     // row_ror:4
  v += P3V_DPP_F32(v, 0x128);     // row_ror:8
  return rows_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, P3V_DPP_F32(v, 0xB1));
  v = fmaxf(v, P3V_DPP_F32(v, 0x4E));
  v = fmaxf(v, P3V_DPP_F32(v, 0x124));
  v = fmaxf(v, P3V_DPP_F32(v, 0x128));
  return rows_max(v);
}

// block-wide sum for blockDim.x = 64*nw (nw <= 16); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

static inline int p3v_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// Row-streaming waves per 4-wave GEMV workgroup: 4, or 3 (the fourth only helps with the prologue) when that spreads the launch's
// `waves` more evenly over the CUs -- 1536 waves as 384 workgroups put two on half of the CUs and one on the others, and the launch
// lasts as long as the doubly loaded ones; as 512 x 3 every CU carries the same (tools/gemv_timeline.py).  forced: tuning knob.
static inline int p3v_gemv_wpw(int waves, int n_cu, int forced) {
  if (forced == 3 || forced == 4) return forced;
  const float w3 = (float)p3v_cdiv(p3v_cdiv(waves, 3), n_cu) * 3.f * n_cu / (float)waves;
  const float w4 = (float)p3v_cdiv(p3v_cdiv(waves, 4), n_cu) * 4.f * n_cu / (float)waves;
  return w3 < w4 - 0.05f ? 3 : 4;
}

// Launch-policy knobs.  Filled ONCE from P3V_<UPPER-CASE NAME> environment variables the first time a launcher asks
// (p3v_runtime.hip), changed afterwards only through p3v_set_tuning() (kernel tests and A/B scripts pin a variant with
// it): no launch reads the environment.  -1 = "let the launcher decide".
struct P3vTuning {
  int gemm_big_rows;        // rows given to the 256x256-tile GEMM (-1: cost model)
  int gemm_no_splitk, gemm_splitk_max_m, gemm_splitk_max_s, gemm_splitk_wgs, gemm_128, gemm_persistent;
  int gemm_rows;            // 1: 9 .. 32 rows on the register-streaming kernel (p3v_gemm_rows.hip), 0: on the 64-row tiles of p3v_gemm_skinny.hip
  int gemm_no_skinny, gemm_skinny_max_m, gemm_skinny_s, gemm_skinny_tm128;   // the 128 x 64-tile weight-streaming kernel for 17 .. max_m rows (p3v_gemm_skinny.hip)
  int gemm_no_qkv_fuse;     // 1: p3v_gemm_qkv reports P3V_ERR_UNSUPPORTED (callers then run p3v_gemm + p3v_rope_kv_append)
  int gemm_f8_narrow;       // -1: by shape, 0 / 1: pin the fp8 tile width
  int attn_no_dma, attn_old, attn_pp, attn_il, attn_il_waves, combine_g, kvq_old, q8_old;
  int attn_fo_map_q8;       // the same placement for the int8-KV twin (k_attn_decode128_q8<true>): 0 = off (measured slower there)
  int attn_fo_map;          // fused decode attention + o_proj: 2 / 1 = roles placed by virtual CU (fo_map, round 6), 0 = the (split, head) grid
  int gemv_no_mfma, gemv_no_mfma8, gemv_wpc, gemv8_wgs, gemv_variant, gemv_rows, gemv8_min, gemv_mfma8, gemv_f8_wpc, gemv_q4_wpc, gemv_wpw, gemv_q4_rows_wgs, gemv_q4_rows8;
};
const P3vTuning& p3v_tuning();
