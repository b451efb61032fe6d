// Micro-benchmark: do MFMA (one wave) and VALU / transcendental work (ANOTHER wave of the same SIMD, or the same wave) overlap
// on gfx950?  The prefill attention kernel's ping-pong design assumes they do.  One workgroup of 512 threads per CU: waves 0-3
// ("group A") and 4-7 ("group B") land pairwise on the four SIMDs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o build/mfma_valu_overlap && build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { W_NONE = 0, W_MFMA16 = 1, W_EXP = 2, W_FMA = 3, W_MFMA32 = 4, W_MIX_EXP = 5, W_MIX_FMA = 6, W_CVT = 7, W_MAX3 = 8, W_PKFMA = 9 };

template <int work>
__device__ __forceinline__ float body(int iters) {
  float r = 0.f;
  if (work == W_MFMA16 || work == W_MIX_EXP || work == W_MIX_FMA) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float e[4] = {threadIdx.x * 1e-3f, threadIdx.x * 1e-3f + 1, threadIdx.x * 1e-3f + 2, threadIdx.x * 1e-3f + 3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        c[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[u & 3], 0, 0, 0);
        if (work == W_MIX_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(e[u & 3]));
        if (work == W_MIX_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(e[u & 1]), "+v"(e[2 + (u & 1)]));
      }
    }
    r = c[0][0] + c[1][1] + c[2][2] + c[3][3] + e[0] + e[1] + e[2] + e[3];
  } else if (work == W_MFMA32) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {                                   // same flops as the 16 x 16 x 32 loop
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      }
    }
    r = c0[0] + c1[1];
  } else if (work != W_NONE) {
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {                                  // 16 independent ops per iteration
        if (work == W_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
        if (work == W_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[i]));
        if (work == W_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(e[i]));
        if (work == W_MAX3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(e[i]));
      }
      if (work == W_PKFMA) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {                             // 8 packed ops = 16 fmas
          float2 t = {e[i], e[i + 1]};
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(t));
          e[i] = t.x, e[i + 1] = t.y;
        }
      }
    }
    for (int i = 0; i < 16; ++i) r += e[i];
  }
  return r;
}

template <int WA, int WB, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r;
  if (wave < 4) {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    r = body<WA>(iters);
  } else {
    r = body<WB>(iters);
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int WA, int WB, int PRIO>
static double run(const char* name, float* out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  k<WA, WB, PRIO><<<256, 512>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a);
    k<WA, WB, PRIO><<<256, 512>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("%-44s %8.1f us   (%.2f ns per loop iteration = 16 MFMAs / 16 VALU ops per wave)\n", name, best * 1e3, best * 1e6 / iters);
  return best;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int it = 20000;
  run<W_MFMA16, W_NONE, 0>("A: mfma 16x16x32       B: idle", out, it);
  run<W_MFMA32, W_NONE, 0>("A: mfma 32x32x16 (x8)  B: idle", out, it);
  run<W_NONE, W_EXP, 0>("A: idle                B: v_exp_f32", out, it);
  run<W_NONE, W_FMA, 0>("A: idle                B: v_fma_f32", out, it);
  run<W_NONE, W_CVT, 0>("A: idle                B: v_cvt_pk_bf16_f32", out, it);
  run<W_NONE, W_MAX3, 0>("A: idle                B: v_max3_f32", out, it);
  run<W_NONE, W_PKFMA, 0>("A: idle                B: v_pk_fma_f32 (8)", out, it);
  run<W_MFMA16, W_EXP, 0>("A: mfma 16x16x32       B: v_exp_f32", out, it);
  run<W_MFMA16, W_EXP, 1>("A: mfma 16x16x32 prio  B: v_exp_f32", out, it);
  run<W_MFMA16, W_FMA, 0>("A: mfma 16x16x32       B: v_fma_f32", out, it);
  run<W_MFMA16, W_FMA, 1>("A: mfma 16x16x32 prio  B: v_fma_f32", out, it);
  run<W_MFMA32, W_EXP, 0>("A: mfma 32x32x16       B: v_exp_f32", out, it);
  run<W_MFMA32, W_FMA, 0>("A: mfma 32x32x16       B: v_fma_f32", out, it);
  run<W_MFMA16, W_PKFMA, 0>("A: mfma 16x16x32       B: v_pk_fma_f32 (8)", out, it);
  run<W_MFMA16, W_CVT, 0>("A: mfma 16x16x32       B: v_cvt_pk_bf16_f32", out, it);
  run<W_MFMA16, W_MFMA16, 0>("A: mfma 16x16x32       B: mfma 16x16x32", out, it);
  run<W_EXP, W_EXP, 0>("A: v_exp_f32           B: v_exp_f32", out, it);
  run<W_FMA, W_FMA, 0>("A: v_fma_f32           B: v_fma_f32", out, it);
  run<W_EXP, W_FMA, 0>("A: v_exp_f32           B: v_fma_f32", out, it);
  run<W_MIX_EXP, W_NONE, 0>("A: mfma + 1 v_exp each (same wave)  B: idle", out, it);
  run<W_MIX_FMA, W_NONE, 0>("A: mfma + 2 v_fma each (same wave)  B: idle", out, it);
  run<W_MIX_EXP, W_MIX_EXP, 0>("A and B: mfma + 1 v_exp each", out, it);
  return 0;
}
