#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned u = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  out[128 + threadIdx.x] = r2[0]; out[192 + threadIdx.x] = r2[1];
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 256 * 4); unsigned h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* n[4] = {"pl32 r0", "pl32 r1", "pl16 r0", "pl16 r1"};
  for (int a = 0; a < 4; ++a) { printf("%s:", n[a]); for (int i = 0; i < 64; ++i) printf(" %u", h[a * 64 + i]); printf("\n"); }
}
