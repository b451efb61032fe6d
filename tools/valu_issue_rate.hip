// Micro-benchmark: VALU issue rate of ONE wave per SIMD against two / four waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_rate.hip -o build/valu_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int NIND>
__global__ void k(float* out, int iters) {
  float e[16];
  for (int i = 0; i < 16; ++i) e[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                    // 16 ops per iteration over NIND independent registers
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[i % NIND]));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i % NIND]));
      if (OP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(e[i % NIND]));
      if (OP == 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(e[i % NIND]));
      if (OP == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(e[i % NIND]));
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += e[i];
  if (r == 123.456f) out[threadIdx.x] = r;
}
template <int OP, int NIND>
static void run(const char* name, float* out, int threads) {
  const int iters = 20000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  k<OP, NIND><<<256, threads>>>(out, iters);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(a);
    k<OP, NIND><<<256, threads>>>(out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double ns_per_op_wave = best * 1e6 / iters / 16;
  printf("%-12s indep %2d  %4d threads/CU (%d waves/SIMD): %6.2f ns per instruction per wave, %6.2f ns per instruction per SIMD\n", name, NIND, threads, threads / 256,
         ns_per_op_wave, ns_per_op_wave / (threads / 256));
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4096);
  for (int t : {256, 512, 1024}) {
    run<0, 16>("v_fma_f32", out, t);
    run<0, 1>("v_fma_f32", out, t);
    run<1, 16>("v_exp_f32", out, t);
    run<1, 1>("v_exp_f32", out, t);
    run<2, 16>("v_max3_f32", out, t);
    run<3, 16>("v_add_f32", out, t);
    run<3, 2>("v_add_f32", out, t);
    run<4, 16>("v_mov_b32", out, t);
  }
  return 0;
}
