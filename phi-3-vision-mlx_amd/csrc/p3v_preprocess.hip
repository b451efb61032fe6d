// On-device image preprocessing: the reference's Phi3VImageProcessor (phi.py:283-372) without its host loops.
//   HD_transform (phi.py:290-310): PIL BILINEAR resize -> white padding to a multiple of 336 rows -> (transpose back for
//   portrait inputs) -> (x / 255 - mean) / std in float64 -> CHW;  crop grid + global view (phi.py:311-372).
// Bit-compatible with the host path (and through it with the reference, tests/golden/ref_processor.*):
//   * the resize is Pillow's ImagingResample for 8-bit pixels (libImaging/Resample.c -- third-party, not in the reference
//     tree): double-precision triangle-filter coefficients normalised per output pixel and rounded to 22-bit fixed point
//     (computed on the host, a few KB), horizontal pass then vertical pass, each `clip8((sum + 2^21) >> 22)`;
//     tests pin the restatement against the installed Pillow on up- and down-scaling cases;
//   * the normalisation is a 256-entry float64 table per channel built with the reference's own expression;
//   * the 336 x 336 global view keeps the reference's arithmetic: fp32 product of the two tap weights, times the float64
//     pixel, summed as 0.0 + ((e00 + e01) + (e10 + e11)) in float64 (see processor.interpolate_336), no FMA contraction.
// Output: float32 = the value the reference's mx.array(float64) cast produces (phi.py:279).
#include "p3v_common.h"

// out[a][xx][i] = clip8((2^21 + sum_x in[a][xmin(xx) + x][i] * k[xx][x]) >> 22): `inner` contiguous bytes per position
__global__ void __launch_bounds__(256) k_resample_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int in_len,
                                                     int out_len, int inner, const int32_t* __restrict__ kk,
                                                     const int32_t* __restrict__ bounds, int ksize) {
  const int a = blockIdx.z, xx = blockIdx.y;
  const int xmin = bounds[2 * xx], xn = bounds[2 * xx + 1];
  const uint8_t* src = in + ((size_t)a * in_len + xmin) * inner;
  uint8_t* dst = out + ((size_t)a * out_len + xx) * inner;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += gridDim.x * blockDim.x) {
    int ss = 1 << 21;
    for (int x = 0; x < xn; ++x) ss += (int)src[(size_t)x * inner + i] * kk[xx * ksize + x];
    ss >>= 22;
    dst[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

extern "C" int p3v_resample_u8(const uint8_t* in, uint8_t* out, int outer, int in_len, int out_len, int inner,
                               const int32_t* coeffs, const int32_t* bounds, int ksize, void* stream) {
  if (!in || !out || !coeffs || !bounds || outer <= 0 || in_len <= 0 || out_len <= 0 || inner <= 0 || ksize <= 0) return P3V_ERR_ARG;
  if (out_len > 65535 || outer > 65535) return P3V_ERR_UNSUPPORTED;
  const dim3 grid(min(64, p3v_cdiv(inner, 256)), out_len, outer);
  hipLaunchKernelGGL(k_resample_u8, grid, dim3(256), 0, (hipStream_t)stream, in, out, in_len, out_len, inner, coeffs, bounds, ksize);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// pixel (Y, X, c) of the HD image = padded (and, for portrait inputs, transposed-back) resized image
struct HdView {
  const uint8_t* img;          // resized image [rh, rw, 3]
  int rh, rw, top, hp, portrait;
};
__device__ __forceinline__ int hd_pixel(const HdView& v, int Y, int X, int c) {
  const int r = v.portrait ? X : Y, col = v.portrait ? Y : X;   // coordinates in the padded landscape image [hp, rw]
  const int rr = r - v.top;
  return (rr < 0 || rr >= v.rh) ? 255 : v.img[((size_t)rr * v.rw + col) * 3 + c];
}

// crop slots 1..: pixel_values[1 + cy * wc + cx][c][y][x] = lut[c][HD(cy*336 + y, cx*336 + x, c)]  (phi.py:313-314)
__global__ void __launch_bounds__(256) k_hd_crops(HdView v, const double* __restrict__ lut, float* __restrict__ pv, int H, int W) {
  const int wc = W / 336;
  const size_t n = (size_t)H * W * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % 336, y = (i / 336) % 336, c = (i / (336 * 336)) % 3, slot = i / (3 * 336 * 336);
    const int cy = slot / wc, cx = slot - cy * wc;
    pv[(size_t)(1 + slot) * (3 * 336 * 336) + ((size_t)c * 336 + y) * 336 + x] = (float)lut[c * 256 + hd_pixel(v, cy * 336 + y, cx * 336 + x, c)];
  }
}

// global view, slot 0 (phi.py:312,331-372): taps 0 and 1 only (taps 2, 3 carry weight 0)
__global__ void __launch_bounds__(256) k_hd_global(HdView v, const double* __restrict__ lut, const float* __restrict__ hw,
                                                   const int32_t* __restrict__ hi, const float* __restrict__ ww,
                                                   const int32_t* __restrict__ wi, float* __restrict__ pv) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 3 * 336 * 336) return;
  const int j = idx % 336, i = (idx / 336) % 336, c = idx / (336 * 336);
  double e[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float wab = __fmul_rn(hw[2 * i + a], ww[2 * j + b]);                       // fp32 x fp32 -> fp32
      e[a][b] = __dmul_rn((double)wab, lut[c * 256 + hd_pixel(v, hi[2 * i + a], wi[2 * j + b], c)]);
    }
  const double s = __dadd_rn(0.0, __dadd_rn(__dadd_rn(e[0][0], e[0][1]), __dadd_rn(e[1][0], e[1][1])));
  pv[idx] = (float)s;
}

extern "C" int p3v_hd_preprocess(const uint8_t* resized, int rh, int rw, int top, int hp, int portrait, const double* lut,
                                 const float* hw, const int32_t* hi, const float* ww, const int32_t* wi, float* pixel_values,
                                 int n_slots, void* stream) {
  if (!resized || !lut || !hw || !hi || !ww || !wi || !pixel_values) return P3V_ERR_ARG;
  if (rh <= 0 || rw <= 0 || top < 0 || hp < rh + top || hp % 336 || rw % 336) return P3V_ERR_ARG;
  const int H = portrait ? rw : hp, W = portrait ? hp : rw;
  const int n_crops = (H / 336) * (W / 336);
  if (n_slots < 1 + n_crops) return P3V_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const HdView v = {resized, rh, rw, top, hp, portrait};
  const size_t slot = (size_t)3 * 336 * 336;
  if (n_slots > 1 + n_crops &&
      hipMemsetAsync(pixel_values + (size_t)(1 + n_crops) * slot, 0, (size_t)(n_slots - 1 - n_crops) * slot * sizeof(float), s) != hipSuccess)
    return P3V_ERR_HIP;
  hipLaunchKernelGGL(k_hd_crops, dim3(2048), dim3(256), 0, s, v, lut, pixel_values, H, W);
  P3V_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_hd_global, dim3(p3v_cdiv(3 * 336 * 336, 256)), dim3(256), 0, s, v, lut, hw, hi, ww, wi, pixel_values);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
