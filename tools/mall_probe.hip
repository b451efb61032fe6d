// Does a second read of the same buffer come from the 256 MiB Infinity Cache, and how fast?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <bool NT>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, size_t n, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  unsigned* out; CK(hipMalloc(&out, 4));
  const size_t pool = (size_t)1024 << 20;
  char* buf; CK(hipMalloc(&buf, pool)); CK(hipMemset(buf, 1, pool));
  for (size_t mb : {19, 50, 100, 200}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    for (int mode = 0; mode < 4; ++mode) {   // 0: cold/plain  1: warm/plain (same buffer)  2: warm, first pass nt then plain  3: warm nt/nt
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 16; ++i) {
        const char* src = buf + (mode == 0 ? (size_t)(i % (pool / bytes)) * bytes : 0);
        if (mode == 3 || (mode == 2 && (i & 1) == 0)) hipLaunchKernelGGL(k_read<true>, dim3(512), dim3(256), 0, s, (const u32x4*)src, n, out);
        else hipLaunchKernelGGL(k_read<false>, dim3(512), dim3(256), 0, s, (const u32x4*)src, n, out);
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const char* names[] = {"cold plain", "warm plain", "warm nt/plain alternating", "warm nt"};
      printf("%4zu MB  %-28s %8.2f us/kernel  %7.1f GB/s\n", mb, names[mode], ms * 1e3 / 16, bytes * 16 / ms / 1e6);
    }
  }
  return 0;
}
