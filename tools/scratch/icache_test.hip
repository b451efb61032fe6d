// Is straight-line code instruction-fetch bound?  One wave per block; N unrolled dependent VALU ops vs the same
// count in a loop; cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void __launch_bounds__(64) straight(float* x, unsigned long long* t) {
  float v = x[threadIdx.x];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v));
  const unsigned long long t1 = __builtin_readcyclecounter();
  x[threadIdx.x] = v;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int N>
__global__ void __launch_bounds__(64) looped(float* x, unsigned long long* t) {
  float v = x[threadIdx.x];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  x[threadIdx.x] = v;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void other(float* x) { x[threadIdx.x] += 1.f; }
int main() {
  float* x; unsigned long long* t; (void)hipMalloc(&x, 4096); (void)hipMalloc(&t, 8 * 1024);
  unsigned long long h[1024];
  auto report = [&](const char* name, int nb) {
    (void)hipMemcpy(h, t, 8 * nb, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull, mx = 0, sum = 0;
    for (int i = 0; i < nb; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
    printf("%-34s blocks %4d: min %6llu  avg %6llu  max %6llu cycles\n", name, nb, mn, sum / nb, mx);
  };
  for (int nb : {1, 256, 768}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(other, dim3(1), dim3(64), 0, 0, x);
      hipLaunchKernelGGL(straight<1024>, dim3(nb), dim3(64), 0, 0, x, t); report("straight 1024 fma (8 KB)", nb);
    }
    hipLaunchKernelGGL(other, dim3(1), dim3(64), 0, 0, x);
    hipLaunchKernelGGL(looped<1024>, dim3(nb), dim3(64), 0, 0, x, t); report("looped 1024 fma (64 x 16)", nb);
    hipLaunchKernelGGL(straight<256>, dim3(nb), dim3(64), 0, 0, x, t); report("straight 256 fma (2 KB)", nb);
    hipLaunchKernelGGL(straight<256>, dim3(nb), dim3(64), 0, 0, x, t); report("straight 256 fma again", nb);
  }
  return 0;
}
