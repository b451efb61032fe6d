// 8 repacked 4-bit weights . 8 bf16 activations: shared by k_gemv3_q4 (p3v_gemv_q4.hip, which documents the layout) and the
// o_proj half of k_attn_decode128_o4 (p3v_attention.hip), which must repeat the GEMV's arithmetic bit for bit.
#pragma once
#include "p3v_common.h"

typedef __bf16 q4_pair_t __attribute__((ext_vector_type(2)));
typedef __bf16 q4_oct_t __attribute__((ext_vector_type(8)));

// 8 weights (one repacked dword) . 8 activations -> += sum_k x_k (128 + q_k)
__device__ __forceinline__ float dot8_q4(uint32_t r, u32x4_t x, float acc) {
  const q4_oct_t xv = __builtin_bit_cast(q4_oct_t, x);
  const uint32_t p0 = (r & 0x000F000Fu) | 0x43004300u, p1 = ((r >> 4) & 0x000F000Fu) | 0x43004300u;
  const uint32_t p2 = ((r >> 8) & 0x000F000Fu) | 0x43004300u, p3 = ((r >> 12) & 0x000F000Fu) | 0x43004300u;
  // (each dword goes through an asm no-op before it is viewed as a bf16 pair: hipcc 7.2 folds such bit_casts, see dot8)
  uint32_t q0 = p0, q1 = p1, q2 = p2, q3 = p3;
  asm("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(q4_pair_t, q0), __builtin_shufflevector(xv, xv, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(q4_pair_t, q1), __builtin_shufflevector(xv, xv, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(q4_pair_t, q2), __builtin_shufflevector(xv, xv, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(q4_pair_t, q3), __builtin_shufflevector(xv, xv, 6, 7), acc, false);
  return acc;
}

