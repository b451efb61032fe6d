// LoRA adapter inference (reference: LoRALinear.__call__, phi.py:129-133)
//   y = linear(x)                      frozen projection, bf16 (any p3v_gemm / p3v_gemv launch, plain epilogue)
//   z = (x @ lora_a) @ lora_b          fp32 (lora_a [K, r], lora_b [r, N] are fp32 in adapters.safetensors)
//   out = (y + scale * z).astype(bf16)
// as two small launches around the frozen projection: p3v_lora_down (t = x @ lora_a, [M, r] fp32) and
// p3v_lora_up (rank-r update of y, fused with the epilogue the frozen projection would have carried).
// Both are bandwidth-trivial next to the projection they decorate (r <= 64): rows of x stream once, lora_a /
// lora_b stay L2-resident.
#include "p3v_common.h"

__global__ void __launch_bounds__(256) k_lora_down(const bf16_t* __restrict__ x, const float* __restrict__ a,
                                                   float* __restrict__ t, int K, int r) {
  __shared__ float red[4][8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* xr = x + (size_t)m * K;
  for (int rc = 0; rc < r; rc += 8) {
    const int nr = min(8, r - rc);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < K; k += 256) {
      const float xv = bf16_to_f32(xr[k]);
      const float* ar = a + (size_t)k * r + rc;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nr) acc[j] += xv * ar[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = wave_sum(acc[j]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][j] = acc[j];
    }
    __syncthreads();
    if (tid < nr) t[(size_t)m * r + rc + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
}

extern "C" int p3v_lora_down(const uint16_t* x, const float* lora_a, float* t, int M, int K, int r, void* stream) {
  if (!x || !lora_a || !t || M < 0 || K <= 0 || r <= 0 || r > 64) return P3V_ERR_ARG;
  if (M == 0) return P3V_OK;
  hipLaunchKernelGGL(k_lora_down, dim3(M), dim3(256), 0, (hipStream_t)stream, x, lora_a, t, K, r);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// MODE 0: out = v;  1: out = bf16(resid + v);  2: out[n] = silu(v[n]) * v[n + N/2] with the per-op bf16 rounding of
// the fused SiLU epilogue (phi.py:469-471).  v = bf16(y + scale * (t @ lora_b)).
template <int MODE>
__global__ void __launch_bounds__(256) k_lora_up(const bf16_t* __restrict__ y, const float* __restrict__ t,
                                                 const float* __restrict__ b, float scale, const bf16_t* __restrict__ resid,
                                                 bf16_t* __restrict__ out, int N, int r) {
  __shared__ float ts[64];
  const int m = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < r) ts[threadIdx.x] = t[(size_t)m * r + threadIdx.x];
  __syncthreads();
  const int n_out = MODE == 2 ? N / 2 : N;
  if (n >= n_out) return;
  auto upd = [&](int col) {
    float z = 0.f;
    for (int j = 0; j < r; ++j) z += ts[j] * b[(size_t)j * N + col];
    return bf16_round(bf16_to_f32(y[(size_t)m * N + col]) + scale * z);
  };
  if (MODE == 2) {
    const float g = upd(n), u = upd(n + n_out);
    out[(size_t)m * n_out + n] = f32_to_bf16(bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * u);
  } else {
    const float v = upd(n);
    out[(size_t)m * N + n] = f32_to_bf16(MODE == 1 ? bf16_to_f32(resid[(size_t)m * N + n]) + v : v);
  }
}

extern "C" int p3v_lora_up(const uint16_t* y, const float* t, const float* lora_b, float scale, int epilogue,
                           const uint16_t* resid, uint16_t* out, int M, int N, int r, void* stream) {
  if (!y || !t || !lora_b || !out || M < 0 || N <= 0 || r <= 0 || r > 64) return P3V_ERR_ARG;
  if (epilogue == P3V_EPI_RESID_BF16 && !resid) return P3V_ERR_ARG;
  if (epilogue == P3V_EPI_SILU_MUL && (N & 1)) return P3V_ERR_ARG;
  if (M == 0) return P3V_OK;
  hipStream_t s = (hipStream_t)stream;
  const int n_out = epilogue == P3V_EPI_SILU_MUL ? N / 2 : N;
  const dim3 grid(p3v_cdiv(n_out, 256), M);
  switch (epilogue) {
    case P3V_EPI_NONE: hipLaunchKernelGGL(k_lora_up<0>, grid, dim3(256), 0, s, y, t, lora_b, scale, resid, out, N, r); break;
    case P3V_EPI_RESID_BF16: hipLaunchKernelGGL(k_lora_up<1>, grid, dim3(256), 0, s, y, t, lora_b, scale, resid, out, N, r); break;
    case P3V_EPI_SILU_MUL: hipLaunchKernelGGL(k_lora_up<2>, grid, dim3(256), 0, s, y, t, lora_b, scale, resid, out, N, r); break;
    default: return P3V_ERR_UNSUPPORTED;
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
