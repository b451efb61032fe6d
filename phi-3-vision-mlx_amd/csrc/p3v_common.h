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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
