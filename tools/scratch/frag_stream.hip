// Which global-load lane patterns can stream a [N, K] bf16 matrix at HBM rate?  (hipcc --offload-arch=gfx950 -O3)
// A: MFMA A-fragment order (lane -> row lane&15, 16 B at k-group lane>>4): every lane quad touches 4 different lines
// B: quad-contiguous (lane -> row lane>>2, 16-byte chunk lane&3): 16 rows x 64 B per instruction
// C: octet-contiguous (lane -> row lane>>3, chunk lane&7): 8 rows x 128 B (whole lines) per instruction
// D: one row, 1 KiB contiguous per instruction (what the M = 1 kernel does)
// E: whole lines like C but lane-scattered: lane -> row lane&7, chunk 4*((lane>>3)&1) + (lane>>4)  (an MFMA A fragment whose
//    16 rows are 8 weight rows x 2 k-halves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ void __launch_bounds__(256) k(const unsigned short* W, int N, int K, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_base = blockIdx.x * 16, kq = K / 4, k_lo = wave * kq;      // 16 rows x K/4 per wave: 24 loads of 1 KiB at K = 3072
  u32x4 v[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const unsigned short* p;
    if (PAT == 0) p = W + (size_t)(n_base + (lane & 15)) * K + k_lo + i * 32 + 8 * (lane >> 4);
    else if (PAT == 1) p = W + (size_t)(n_base + (lane >> 2)) * K + k_lo + i * 32 + 8 * (lane & 3);
    else if (PAT == 2) p = W + (size_t)(n_base + (i & 1) * 8 + (lane >> 3)) * K + k_lo + (i >> 1) * 64 + 8 * (lane & 7);
    else if (PAT == 5) p = W + (size_t)(n_base + (i & 1) * 8 + ((lane & 15) >> 1)) * K + k_lo + (i >> 1) * 64 + 8 * (2 * (lane >> 4) + (lane & 1));   // F: pairs
    else if (PAT == 4) p = W + (size_t)(n_base + (i & 1) * 8 + (lane & 7)) * K + k_lo + (i >> 1) * 64 + 8 * (4 * ((lane >> 3) & 1) + (lane >> 4));
    else p = W + (size_t)(n_base + (i & 15)) * K + k_lo + (i >> 4) * 512 + 8 * lane;     // rows 0..15 x 512 k, then 8 more rows' second half
    v[i] = __builtin_nontemporal_load((const u32x4*)p);
  }
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  if (acc == 0x12345678u) sink[0] = 1.f;
}
int main() {
  const int N = 9216, K = 3072, NB = 6;
  unsigned short* W; float* sink;
  hipMalloc(&W, (size_t)NB * N * K * 2); hipMalloc(&sink, 4);
  hipMemset(W, 1, (size_t)NB * N * K * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int pat = 0; pat < 6; ++pat) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      for (int i = 0; i < 12; ++i) {
        const unsigned short* w = W + (size_t)(i % NB) * N * K;
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
        if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
        if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
        if (pat == 5) hipLaunchKernelGGL(k<5>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
        if (pat == 4) hipLaunchKernelGGL(k<4>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
        if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(N / 16), dim3(256), 0, 0, w, N, K, sink);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms / 12 < best) best = ms / 12;
    }
    printf("pattern %c: %.2f us per 56.6 MB launch (incl. launch gap) = %.2f TB/s\n", "ABCDEF"[pat], best * 1e3, (double)N * K * 2 / best / 1e9);
  }
  return 0;
}
