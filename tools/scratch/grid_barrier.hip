// Latency of a software grid barrier on MI355X (persistent kernel, all blocks resident), with bounded spinning.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { *err = 1; ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}
__global__ void __launch_bounds__(256) k(unsigned* cnt, unsigned* err, float* data, int iters, int work) {
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc += data[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xfffff];
    if (!grid_barrier(cnt, (unsigned)(it + 1) * gridDim.x, err)) break;
  }
  if (acc == 12345.f) data[0] = acc;
}
int main(int argc, char** argv) {
  unsigned *cnt, *err; float* data;
  (void)hipMalloc(&cnt, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&data, 4 << 20);
  (void)hipMemset(data, 0, 4 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nb : {256, 512, 1024}) for (int work : {0, 4}) {
    const int iters = 2000;
    (void)hipMemset(cnt, 0, 4); (void)hipMemset(err, 0, 4);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, cnt, err, data, iters, work);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned herr; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("blocks %4d work %d: %.3f us per barrier%s\n", nb, work, ms * 1e3 / iters, herr ? "  (TIMEOUT)" : "");
  }
  return 0;
}
