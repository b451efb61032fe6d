// Cold streaming-read rate vs buffer size (graph of 16 launches over rotating buffers > 256 MiB total).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int UN>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, size_t n, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + (UN - 1) * stride < n; i += UN * stride) {
    u32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < UN; ++u) acc ^= v[u];
  }
  for (; i < n; i += stride) acc ^= p[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  unsigned* out; CK(hipMalloc(&out, 4));
  const size_t pool = (size_t)1600 << 20;
  char* buf; CK(hipMalloc(&buf, pool)); CK(hipMemset(buf, 1, pool));
  for (size_t mb : {19, 50, 57, 100, 200, 400, 800}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    const int nrot = (int)(pool / bytes);
    for (int blocks_per_cu : {2, 4, 8}) {
      const int grid = 256 * blocks_per_cu;
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, s, (const u32x4*)(buf + (size_t)(i % nrot) * bytes), n, out);
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%4zu MB  grid %5d (x256 thr)  %8.2f us/kernel  %7.1f GB/s\n", mb, grid, ms * 1e3 / 16, bytes * 16 / ms / 1e6);
    }
  }
  return 0;
}
