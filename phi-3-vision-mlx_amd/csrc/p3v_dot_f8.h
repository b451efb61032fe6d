// 16 e4m3 weights . 16 bf16 activations -> fp32: shared by k_gemv3_f8 (p3v_gemv_fp8.hip) and the o_proj half of
// k_attn_decode128_q8_o (p3v_attention.hip), which must repeat the GEMV's arithmetic bit for bit.
#pragma once
#include "p3v_common.h"

// 16 fp8 weights (one 16-byte chunk) . 16 bf16 activations (two 16-byte chunks)
// 16 e4m3 weights x 16 bf16 activations: every fp8 PAIR becomes a bf16 pair in one v_cvt_scalef32_pk_bf16_fp8 (exact:
// 3 mantissa bits fit in 7) and meets its two activations in one v_dot2c_f32_bf16 -- 2 VALU instructions per 2 weights
// instead of 1 convert + 2 unpacks + 2 FMAs.  The activation pairs come from an 8 x bf16 view (see dot8 in p3v_gemv.hip).
typedef __bf16 f8_bf16pair_t __attribute__((ext_vector_type(2)));
typedef __bf16 f8_bf16oct_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float dot16_f8(u32x4_t w, u32x4_t xa, u32x4_t xb, float acc) {
  const f8_bf16oct_t a = __builtin_bit_cast(f8_bf16oct_t, xa), b = __builtin_bit_cast(f8_bf16oct_t, xb);
#define P3V_F8_STEP(q, xv, i0)                                                                                           \
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[q], 1.0f, false),                    \
                                        __builtin_shufflevector(xv, xv, i0, i0 + 1), acc, false);                        \
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[q], 1.0f, true),                     \
                                        __builtin_shufflevector(xv, xv, i0 + 2, i0 + 3), acc, false);
  P3V_F8_STEP(0, a, 0) P3V_F8_STEP(1, a, 4) P3V_F8_STEP(2, b, 0) P3V_F8_STEP(3, b, 4)
#undef P3V_F8_STEP
  return acc;
}

