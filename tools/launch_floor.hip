// Measures the per-kernel boundary cost on this box: eager C loop vs hipGraph (stream capture) vs explicit graph nodes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void k_tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f; }
__global__ void k_256(float* x) { x[blockIdx.x * 256 + threadIdx.x] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  float* d; CK(hipMalloc(&d, 1 << 22)); CK(hipMemset(d, 0, 1 << 22));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 200;
  for (int variant = 0; variant < 2; ++variant) {
    auto launch = [&]() { if (variant == 0) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d); else hipLaunchKernelGGL(k_256, dim3(1024), dim3(256), 0, s, d); };
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::high_resolution_clock::now();
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("variant %d eager: %.2f us/kernel (gpu events), %.2f us/kernel (host wall)\n", variant, ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 3; ++rep) {
      t0 = std::chrono::high_resolution_clock::now();
      CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
      t1 = std::chrono::high_resolution_clock::now();
      CK(hipEventElapsedTime(&ms, a, b));
      printf("variant %d graph(capture): %.2f us/kernel (gpu events), %.2f (host wall)\n", variant, ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
