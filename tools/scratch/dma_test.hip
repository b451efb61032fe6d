#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef const __attribute__((address_space(1))) void* gp;
typedef __attribute__((address_space(3))) void* lp;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k(const unsigned char* src, unsigned char* dst, int mode) {
  __shared__ __attribute__((aligned(1024))) unsigned char A[12288];
  __shared__ __attribute__((aligned(1024))) unsigned char Bv[12288];
  const int lane = threadIdx.x;
  for (int i = lane; i < 12288 / 4; i += 64) { ((unsigned*)A)[i] = 0xdeadbeef; ((unsigned*)Bv)[i] = 0xdeadbeef; }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 12; ++it)
    __builtin_amdgcn_global_load_lds((gp)(src + it * 1024 + lane * 16), (lp)(A + it * 1024), 16, 0, 0);
#pragma unroll
  for (int it = 0; it < 12; ++it)
    __builtin_amdgcn_global_load_lds((gp)(src + 12288 + it * 1024 + lane * 16), (lp)(Bv + it * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (mode) __syncthreads();
  for (int i = lane; i < 12288 / 16; i += 64) {
    ((u32x4*)dst)[i] = ((u32x4*)A)[i];
    ((u32x4*)dst)[768 + i] = ((u32x4*)Bv)[i];
  }
}
int main() {
  std::vector<unsigned char> h(24576), o(24576);
  for (int i = 0; i < 24576; ++i) h[i] = (unsigned char)((i * 7 + i / 256) & 0xff);
  unsigned char *s, *d; hipMalloc(&s, 24576); hipMalloc(&d, 24576);
  hipMemcpy(s, h.data(), 24576, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(d, 0, 24576);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, mode);
    hipMemcpy(o.data(), d, 24576, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < 24576; ++i) if (o[i] != h[i]) { if (first < 0) first = i; ++bad; }
    printf("mode %d: %d bad bytes, first at %d\n", mode, bad, first);
  }
  return 0;
}
