// HBM-bound row kernels of the hot path: gather, norms, RoPE/KV append, HD merge,
// patch unfold, argmax / log-softmax / top-k.  All loads are 16-byte vectors
// (8 bf16 or 4 f32 per lane), reductions use 64-lane shuffles.
#include "p3v_common.h"

// ---------------------------------------------------------------- embed gather
__global__ void __launch_bounds__(128) k_embed_gather(const int32_t* __restrict__ ids, const u32x4_t* __restrict__ table,
                                                      u32x4_t* __restrict__ out, int chunks, int vocab) {
  const int t = blockIdx.x;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4_t* src = table + (size_t)id * chunks;
  u32x4_t* dst = out + (size_t)t * chunks;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) dst[c] = src[c];
}

extern "C" int p3v_embed_gather(const int32_t* ids, const uint16_t* table, uint16_t* out, int n_tok, int hidden, int vocab,
                                void* stream) {
  if (!ids || !table || !out || n_tok < 0 || hidden % 8 || vocab <= 0) return P3V_ERR_ARG;
  if (n_tok == 0) return P3V_OK;
  hipLaunchKernelGGL(k_embed_gather, dim3(n_tok), dim3(128), 0, (hipStream_t)stream, ids, (const u32x4_t*)table,
                     (u32x4_t*)out, hidden / 8, vocab);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- RMSNorm (one wave per row)
__global__ void __launch_bounds__(256) k_rmsnorm(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ w,
                                                 u32x4_t* __restrict__ y, int rows, int chunks, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const u32x4_t* xr = x + (size_t)row * chunks;
  float ss = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    u32x4_t v = xr[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bf16lo(v[j]), b = bf16hi(v[j]);
      ss += a * a + b * b;
    }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss * inv_h + eps);
  u32x4_t* yr = y + (size_t)row * chunks;
  for (int c = lane; c < chunks; c += 64) {
    u32x4_t v = xr[c], g = w[c], o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = rms_pair(v[j], r, g[j]);
    yr[c] = o;
  }
}

// the same with the row held in registers (CH 16-byte chunks per lane): x is read once instead of twice
template <int CH>
__global__ void __launch_bounds__(256) k_rmsnorm_r(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ w,
                                                   u32x4_t* __restrict__ y, int rows, int chunks, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const u32x4_t* xr = x + (size_t)row * chunks;
  u32x4_t v[CH], g[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < chunks ? xr[c] : (u32x4_t){0u, 0u, 0u, 0u};
    g[i] = c < chunks ? w[c] : (u32x4_t){0u, 0u, 0u, 0u};
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16lo(v[i][j]), b = bf16hi(v[i][j]);
      ss += a * a + b * b;
    }
  const float r = rsqrtf(wave_sum(ss) * inv_h + eps);
  u32x4_t* yr = y + (size_t)row * chunks;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    if (c < chunks) {
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rms_pair(v[i][j], r, g[i][j]);
      yr[c] = o;
    }
  }
}

extern "C" int p3v_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int hidden, float eps, void* stream) {
  if (!x || !w || !y || rows < 0 || hidden <= 0 || hidden % 8) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  const int chunks = hidden / 8;
  if (chunks <= 6 * 64) {   // hidden <= 3072: same per-lane summation order as k_rmsnorm, so the results are bit-identical
    hipLaunchKernelGGL(k_rmsnorm_r<6>, dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x,
                       (const u32x4_t*)w, (u32x4_t*)y, rows, chunks, 1.0f / hidden, eps);
    P3V_CHECK_LAUNCH();
    return P3V_OK;
  }
  hipLaunchKernelGGL(k_rmsnorm, dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)x,
                     (const u32x4_t*)w, (u32x4_t*)y, rows, hidden / 8, 1.0f / hidden, eps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- LayerNorm f32 -> bf16 (one wave per row)
template <bool OUT_F32>
__global__ void __launch_bounds__(256) k_layernorm(const float4* __restrict__ x, const u32x2_t* __restrict__ w,
                                                   const u32x2_t* __restrict__ b, void* __restrict__ yv, int rows,
                                                   int chunks, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = x + (size_t)row * chunks;
  float s = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    float4 v = xr[c];
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mu = wave_sum(s) * inv_h;
  float q = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    float4 v = xr[c];
    float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float r = rsqrtf(wave_sum(q) * inv_h + eps);
  for (int c = lane; c < chunks; c += 64) {
    float4 v = xr[c];
    u32x2_t g = w[c], be = b[c];
    float4 t;
    t.x = (v.x - mu) * r * bf16lo(g[0]) + bf16lo(be[0]);
    t.y = (v.y - mu) * r * bf16hi(g[0]) + bf16hi(be[0]);
    t.z = (v.z - mu) * r * bf16lo(g[1]) + bf16lo(be[1]);
    t.w = (v.w - mu) * r * bf16hi(g[1]) + bf16hi(be[1]);
    if (OUT_F32) {
      ((float4*)yv)[(size_t)row * chunks + c] = t;
    } else {
      u32x2_t o;
      o[0] = pack_bf16x2(t.x, t.y);
      o[1] = pack_bf16x2(t.z, t.w);
      ((u32x2_t*)yv)[(size_t)row * chunks + c] = o;
    }
  }
}

// the same with the row held in registers (CH float4 per lane): x is read once instead of three times
template <bool OUT_F32, int CH>
__global__ void __launch_bounds__(256) k_layernorm_r(const float4* __restrict__ x, const u32x2_t* __restrict__ w,
                                                     const u32x2_t* __restrict__ b, void* __restrict__ yv, int rows,
                                                     int chunks, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = x + (size_t)row * chunks;
  float4 v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < chunks ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mu = wave_sum(s) * inv_h;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    if (i * 64 + lane < chunks) {
      const float d0 = v[i].x - mu, d1 = v[i].y - mu, d2 = v[i].z - mu, d3 = v[i].w - mu;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float r = rsqrtf(wave_sum(q) * inv_h + eps);
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    if (c < chunks) {
      const u32x2_t g = w[c], be = b[c];
      float4 t;
      t.x = (v[i].x - mu) * r * bf16lo(g[0]) + bf16lo(be[0]);
      t.y = (v[i].y - mu) * r * bf16hi(g[0]) + bf16hi(be[0]);
      t.z = (v[i].z - mu) * r * bf16lo(g[1]) + bf16lo(be[1]);
      t.w = (v[i].w - mu) * r * bf16hi(g[1]) + bf16hi(be[1]);
      if (OUT_F32) {
        ((float4*)yv)[(size_t)row * chunks + c] = t;
      } else {
        u32x2_t o;
        o[0] = pack_bf16x2(t.x, t.y);
        o[1] = pack_bf16x2(t.z, t.w);
        ((u32x2_t*)yv)[(size_t)row * chunks + c] = o;
      }
    }
  }
}

extern "C" int p3v_layernorm(const float* x, const uint16_t* w, const uint16_t* b, void* y, int out_f32, int rows,
                             int hidden, float eps, void* stream) {
  if (!x || !w || !b || !y || rows < 0 || hidden <= 0 || hidden % 4) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  if (hidden / 4 <= 4 * 64) {
    if (out_f32)
      hipLaunchKernelGGL((k_layernorm_r<true, 4>), dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                         (const u32x2_t*)w, (const u32x2_t*)b, y, rows, hidden / 4, 1.0f / hidden, eps);
    else
      hipLaunchKernelGGL((k_layernorm_r<false, 4>), dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                         (const u32x2_t*)w, (const u32x2_t*)b, y, rows, hidden / 4, 1.0f / hidden, eps);
    P3V_CHECK_LAUNCH();
    return P3V_OK;
  }
  if (out_f32)
    hipLaunchKernelGGL(k_layernorm<true>, dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const u32x2_t*)w, (const u32x2_t*)b, y, rows, hidden / 4, 1.0f / hidden, eps);
  else
    hipLaunchKernelGGL(k_layernorm<false>, dim3(p3v_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const u32x2_t*)w, (const u32x2_t*)b, y, rows, hidden / 4, 1.0f / hidden, eps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- SuRoPE tables
__global__ void k_rope_table(const float* __restrict__ pos, const float* __restrict__ inv_freq, float scale,
                             float* __restrict__ c, float* __restrict__ s, int n_pos, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pos * half) return;
  const float e = pos[i / half] * inv_freq[i % half];
  c[i] = cosf(e) * scale;
  s[i] = sinf(e) * scale;
}

extern "C" int p3v_rope_table(const float* pos, const float* inv_freq, float scale, float* cos_out, float* sin_out,
                              int n_pos, int half_dim, void* stream) {
  if (!pos || !inv_freq || !cos_out || !sin_out || n_pos < 0 || half_dim <= 0) return P3V_ERR_ARG;
  if (n_pos == 0) return P3V_OK;
  const long n = (long)n_pos * half_dim;
  hipLaunchKernelGGL(k_rope_table, dim3(p3v_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, inv_freq, scale,
                     cos_out, sin_out, n_pos, half_dim);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- split + RoPE + KV append
// one block per token (b,l); work item = 8 consecutive dims of the low half paired with the
// same 8 dims of the high half (q and k heads), or one 8-wide chunk of v.
__device__ __forceinline__ void rope_kv_append_token(int tok, const bf16_t* __restrict__ qkv, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, bf16_t* __restrict__ q_out,
                                                    bf16_t* __restrict__ k_dst, bf16_t* __restrict__ v_dst, int L,
                                                    int nh, int nkv, int hd, int past, const int32_t* d_past,
                                                    int dst_t, int dst_off_is_past, int tab_t, int tab_div, int skip_v, float q_scale) {
  const int b = tok / L, l = tok % L;
  if (d_past) past = *d_past;
  const int half = hd >> 1, hc = half >> 3;          // 8-wide chunks per half
  const int pos = past + l;
  const bool rot = cos_t != nullptr;          // null tables: plain head split (CLIP q/k/v, phi.py:147)
  const float* ct = rot ? cos_t + ((size_t)(b / tab_div) * tab_t + pos) * half : nullptr;
  const float* st = rot ? sin_t + ((size_t)(b / tab_div) * tab_t + pos) * half : nullptr;
  const bf16_t* row = qkv + (size_t)tok * (nh + 2 * nkv) * hd;
  const int dpos = (dst_off_is_past ? past : 0) + l;
  const int n_rot = (nh + nkv) * hc;
  const int n_v = skip_v ? 0 : nkv * (hd >> 3);
  for (int it = threadIdx.x; it < n_rot + n_v; it += blockDim.x) {
    if (it < n_rot) {
      const int head = it / hc, c = it % hc;
      const float qs = head < nh ? q_scale : 1.f;           // queries leave pre-scaled (one rounding, after the multiply)
      const bf16_t* src = row + (size_t)head * hd + c * 8;
      u32x4_t lo = *(const u32x4_t*)src, hi = *(const u32x4_t*)(src + half);
      u32x4_t olo = lo, ohi = hi;
      float cs[8], sn[8];
      if (rot) {
        const float4 c0 = *(const float4*)(ct + c * 8), c1 = *(const float4*)(ct + c * 8 + 4);
        const float4 s0 = *(const float4*)(st + c * 8), s1 = *(const float4*)(st + c * 8 + 4);
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
      }
#pragma unroll
      for (int j = 0; j < 4 && rot; ++j) {
        const float a0 = bf16lo(lo[j]), a1 = bf16hi(lo[j]), b0 = bf16lo(hi[j]), b1 = bf16hi(hi[j]);
        float x0, y0, x1, y1;
        p3v_rope_pair(a0, b0, cs[2 * j], sn[2 * j], qs, x0, y0);
        p3v_rope_pair(a1, b1, cs[2 * j + 1], sn[2 * j + 1], qs, x1, y1);
        olo[j] = pack_bf16x2(x0, x1);
        ohi[j] = pack_bf16x2(y0, y1);
      }
      if (!rot && qs != 1.f) {                                // plain head split (CLIP) with pre-scaled queries
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          olo[j] = pack_bf16x2(bf16lo(lo[j]) * qs, bf16hi(lo[j]) * qs);
          ohi[j] = pack_bf16x2(bf16lo(hi[j]) * qs, bf16hi(hi[j]) * qs);
        }
      }
      bf16_t* dst;
      if (head < nh) dst = q_out + (((size_t)b * nh + head) * L + l) * hd + c * 8;
      else dst = k_dst + (((size_t)b * nkv + (head - nh)) * dst_t + dpos) * hd + c * 8;
      *(u32x4_t*)dst = olo;
      *(u32x4_t*)(dst + half) = ohi;
    } else {
      const int j = it - n_rot, head = j / (hd >> 3), c = j % (hd >> 3);
      const u32x4_t v = *(const u32x4_t*)(row + (size_t)(nh + nkv + head) * hd + c * 8);
      bf16_t* vd = v_dst + (((size_t)b * nkv + head) * hd + c * 8) * (size_t)dst_t + dpos;    // V^T: [hd][dst_t]
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        vd[(size_t)(2 * jj) * dst_t] = (bf16_t)(v[jj] & 0xffff);
        vd[(size_t)(2 * jj + 1) * dst_t] = (bf16_t)(v[jj] >> 16);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_rope_kv_append(const bf16_t* __restrict__ qkv, const float* __restrict__ cos_t,
                                                        const float* __restrict__ sin_t, bf16_t* __restrict__ q_out,
                                                        bf16_t* __restrict__ k_dst, bf16_t* __restrict__ v_dst, int L,
                                                        int nh, int nkv, int hd, int past, const int32_t* d_past,
                                                        int dst_t, int dst_off_is_past, int tab_t, int tab_div, int skip_v, float q_scale) {
  rope_kv_append_token(blockIdx.x, qkv, cos_t, sin_t, q_out, k_dst, v_dst, L, nh, nkv, hd, past, d_past, dst_t, dst_off_is_past, tab_t,
                       tab_div, skip_v, q_scale);
}

// V rows of 64 tokens x one kv head -> V^T columns, transposed through LDS so that both the qkv reads (192 B
// per token) and the cache writes (128 B per d row) are full-line vector accesses.
__device__ __forceinline__ void v_transpose_append_tile(int xb, int head, int b, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ v_dst,
                                                        int L, int nh, int nkv, int hd, int past, const int32_t* d_past, int dst_t,
                                                        int dst_off_is_past) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * (96 + 8)];
  const int ld = hd + 8, cpr = hd >> 3;
  const int l0 = xb * 64;
  if (d_past) past = *d_past;
  const int row_w = (nh + 2 * nkv) * hd;
  for (int i = threadIdx.x; i < 64 * cpr; i += 256) {
    const int tok = i / cpr, c = i % cpr, l = l0 + tok;
    u32x4_t v = {0, 0, 0, 0};
    if (l < L) v = *(const u32x4_t*)(qkv + ((size_t)b * L + l) * row_w + (size_t)(nh + nkv + head) * hd + c * 8);
    *(u32x4_t*)(tile + tok * ld + c * 8) = v;
  }
  __syncthreads();
  const int dpos0 = (dst_off_is_past ? past : 0) + l0;
  bf16_t* base = v_dst + ((size_t)b * nkv + head) * (size_t)hd * dst_t + dpos0;
  const bool aligned = (dpos0 & 7) == 0;
  for (int i = threadIdx.x; i < hd * 8; i += 256) {
    const int d = i >> 3, c = i & 7, n_ok = min(8, L - (l0 + c * 8));
    if (n_ok <= 0) continue;
    bf16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = tile[(c * 8 + j) * ld + d];
    bf16_t* dst = base + (size_t)d * dst_t + c * 8;
    if (aligned && n_ok == 8) {
      u32x4_t w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
      *(u32x4_t*)dst = w;
    } else {
      for (int j = 0; j < n_ok; ++j) dst[j] = e[j];
    }
  }
}

// Prompt-sized calls: ONE launch for both -- the first B * L workgroups split / rotate one token each, the remaining
// ceil(L / 64) * nkv * B transpose one 64-token V tile each (two launches before: a kernel boundary is ~2.4 us, and the two
// memory streams overlap).
__global__ void __launch_bounds__(256) k_rope_kv_vt(const bf16_t* __restrict__ qkv, const float* __restrict__ cos_t,
                                                    const float* __restrict__ sin_t, bf16_t* __restrict__ q_out,
                                                    bf16_t* __restrict__ k_dst, bf16_t* __restrict__ v_dst, int B, int L,
                                                    int nh, int nkv, int hd, int past, const int32_t* d_past,
                                                    int dst_t, int dst_off_is_past, int tab_t, int tab_div, float q_scale) {
  const int n_tok = B * L;
  if ((int)blockIdx.x < n_tok) {
    rope_kv_append_token(blockIdx.x, qkv, cos_t, sin_t, q_out, k_dst, v_dst, L, nh, nkv, hd, past, d_past, dst_t, dst_off_is_past, tab_t,
                         tab_div, 1, q_scale);
  } else {
    const int t = blockIdx.x - n_tok, nx = (L + 63) >> 6;
    v_transpose_append_tile(t % nx, (t / nx) % nkv, t / (nx * nkv), qkv, v_dst, L, nh, nkv, hd, past, d_past, dst_t, dst_off_is_past);
  }
}

extern "C" int p3v_rope_kv_append(const uint16_t* qkv, const float* cos_t, const float* sin_t, uint16_t* q_out,
                                  uint16_t* k_dst, uint16_t* v_dst, int B, int L, int n_heads, int n_kv, int hd, int past,
                                  const int32_t* d_past, int dst_t, int dst_off_is_past, int tab_t, int tab_div,
                                  float q_scale, void* stream) {
  if (!qkv || !q_out || !k_dst || !v_dst || (!cos_t) != (!sin_t)) return P3V_ERR_ARG;
  if (!(q_scale > 0.f)) return P3V_ERR_ARG;
  if (B < 0 || L < 0 || hd % 16 || hd > 96 || n_heads <= 0 || n_kv <= 0 || tab_div <= 0) return P3V_ERR_ARG;
  if (B * L == 0) return P3V_OK;
  const int bulk_v = L >= 32;                             // prefill-shaped: V goes through the LDS transpose kernel
  if (bulk_v) {
    hipLaunchKernelGGL(k_rope_kv_vt, dim3(B * L + p3v_cdiv(L, 64) * n_kv * B), dim3(256), 0, (hipStream_t)stream, qkv, cos_t, sin_t, q_out,
                       k_dst, v_dst, B, L, n_heads, n_kv, hd, past, d_past, dst_t, dst_off_is_past, tab_t, tab_div, q_scale);
  } else {
    hipLaunchKernelGGL(k_rope_kv_append, dim3(B * L), dim3(256), 0, (hipStream_t)stream, qkv, cos_t, sin_t, q_out, k_dst,
                       v_dst, L, n_heads, n_kv, hd, past, d_past, dst_t, dst_off_is_past, tab_t, tab_div, 0, q_scale);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- MLX 4-bit prompt cache (quantize_cache=True, cache_format="mlx4")
// The reference's quantised KV cache (phi.py:528-540): on the FIRST call the keys / values of every (batch row, kv head) are
// flattened to one row of S * hd values and mx.quantize'd with group_size 32, 4 bits -- i.e. every token's 96 dims are three
// groups -- and every later call attends on mx.dequantize of that (later tokens stay unquantised).  One thread per (row, token,
// group): the MLX affine group quantiser (weights.mlx_quantize: the larger-magnitude end of the range is represented exactly,
// scale and bias stay fp32 as they do for the reference's fp32 keys), the codes in MLX's packing (code k of a word at bits
// [4k, 4k+4)), and the group written BACK dequantised (scale * q, + bias, then the cache's bf16; V: every primitive of the composite
// rounds to bf16, as mlx 0.15.0 computes it for a bf16 input): the cache rows then hold exactly
// what the reference attends on from the second call on, and the decode kernels read them as they are.
struct Mlx4Src {                                                  // where the EXACT fp32 keys come from (null qkv: from the bf16 cache rows)
  const bf16_t* qkv; const float* cos_t; const float* sin_t;      // the layer's projection output [B * L, (nh + 2 nkv) * hd] + rotation tables
  int L, nh, nkv, past, tab_t, tab_div;
};

__global__ void __launch_bounds__(256) k_kv_quantize_mlx4(bf16_t* __restrict__ k, bf16_t* __restrict__ vt, uint32_t* __restrict__ k4,
                                                          uint32_t* __restrict__ v4, float* __restrict__ k_sb, float* __restrict__ v_sb,
                                                          int hd, int cache_t, int n_tok, long total, Mlx4Src src) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int gpt = hd >> 5;                                        // groups per token (3)
  const int t = (int)(i % n_tok);                                 // token fastest: the V^T accesses of a wave are then contiguous
  const long r = i / n_tok;
  const int g = (int)(r % gpt);
  const long row2 = r / gpt;                                      // [K rows | V rows]
  const long BH = total / ((long)n_tok * gpt * 2);
  const bool is_v = row2 >= BH;
  const long row = is_v ? row2 - BH : row2;
  float w[32];
  if (!is_v && src.qkv) {
    // The reference quantises its fp32 keys (phi.py:451, 531: RoPE promotes k to fp32 and the cache takes it as it is); the cache row
    // holds them rounded to bf16 -- 0.4 % away, enough to flip 4-bit codes.  Recompute the rotation from the projection output with
    // the arithmetic the rotation kernels use (p3v_rope_pair: multiply, multiply, add in fp32 = the reference's expression).
    const int b = (int)(row / src.nkv), head = (int)(row % src.nkv), half = hd >> 1;
    const bf16_t* x = src.qkv + ((size_t)b * src.L + t) * (size_t)((src.nh + 2 * src.nkv) * hd) + (size_t)(src.nh + head) * hd;
    const size_t trow = ((size_t)(b / src.tab_div) * src.tab_t + src.past + t) * half;
#pragma unroll
    for (int d = 0; d < 32; ++d) {
      const int dd = g * 32 + d, pr = dd < half ? dd : dd - half;
      float o1, o2;
      p3v_rope_pair(bf16_to_f32(x[pr]), bf16_to_f32(x[pr + half]), src.cos_t[trow + pr], src.sin_t[trow + pr], 1.f, o1, o2);
      w[d] = dd < half ? o1 : o2;
    }
  } else if (!is_v) {
    const u32x4_t* srcp = (const u32x4_t*)(k + ((size_t)row * cache_t + t) * hd + g * 32);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4_t v = srcp[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) { w[c * 8 + 2 * j] = bf16lo(v[j]); w[c * 8 + 2 * j + 1] = bf16hi(v[j]); }
    }
  } else {
#pragma unroll
    for (int d = 0; d < 32; ++d) w[d] = bf16_to_f32(vt[((size_t)row * hd + g * 32 + d) * cache_t + t]);
  }
  // mlx 0.15.0's mx.quantize / mx.dequantize are composites of array primitives in the INPUT's dtype (weights.mlx_quantize): the
  // reference's values reach the cache as bf16 (phi.py:443-449: only q and k are promoted by the rotation), so for V every
  // intermediate -- range, scale, edge / scale, edge / q0, w - bias, (w - bias) / scale, scale * q, + bias -- rounds to bf16 (R);
  // its keys are fp32 arrays and see plain fp32 arithmetic.
#define R(x) (is_v ? bf16_round(x) : (x))
  float w_max = w[0], w_min = w[0];
#pragma unroll
  for (int d = 1; d < 32; ++d) { w_max = fmaxf(w_max, w[d]); w_min = fminf(w_min, w[d]); }
  const bool mask = fabsf(w_min) > fabsf(w_max);
  float scale = fmaxf(R(R(w_max - w_min) / 15.f), R(1e-7f));
  scale = mask ? scale : -scale;
  const float edge = mask ? w_min : w_max;
  const float q0 = rintf(R(edge / scale));
  if (q0 != 0.f) scale = R(edge / q0);
  const float bias = q0 == 0.f ? 0.f : edge;
  uint32_t words[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int d = 0; d < 32; ++d) {
    float diff = w[d] - bias;                             // (the empty asm statements keep hipcc, fp-contract=fast, from fusing two
    asm volatile("" : "+v"(diff));                        // primitives into one fma or reassociating across a rounding)
    const float q = fminf(fmaxf(rintf(R(R(diff) / scale)), 0.f), 15.f);
    words[d >> 3] |= (uint32_t)q << (4 * (d & 7));
    float prod = scale * q;                               // two roundings, as mx.dequantize's multiply-then-add
    asm volatile("" : "+v"(prod));
    w[d] = R(prod) + bias;
  }
#undef R
  uint32_t* c4 = (is_v ? v4 : k4) + (((size_t)row * n_tok + t) * gpt + g) * 4;
  *(u32x4_t*)c4 = (u32x4_t){words[0], words[1], words[2], words[3]};
  float* sb = (is_v ? v_sb : k_sb) + (((size_t)row * n_tok + t) * gpt + g) * 2;
  sb[0] = scale, sb[1] = bias;
  if (!is_v) {
    u32x4_t* dst = (u32x4_t*)(k + ((size_t)row * cache_t + t) * hd + g * 32);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      u32x4_t v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(w[c * 8 + 2 * j], w[c * 8 + 2 * j + 1]);
      dst[c] = v;
    }
  } else {
#pragma unroll
    for (int d = 0; d < 32; ++d) vt[((size_t)row * hd + g * 32 + d) * cache_t + t] = f32_to_bf16(w[d]);
  }
}

extern "C" int p3v_kv_quantize_mlx4(uint16_t* k, uint16_t* vt, uint32_t* k4, uint32_t* v4, float* k_sb, float* v_sb, int BH, int hd,
                                    int cache_t, int n_tok, const uint16_t* qkv, const float* cos_t, const float* sin_t, int n_heads,
                                    int n_kv, int past, int tab_t, int tab_div, void* stream) {
  if (!k || !vt || !k4 || !v4 || !k_sb || !v_sb) return P3V_ERR_ARG;
  if (BH < 0 || n_tok < 0 || hd <= 0 || hd % 32 || n_tok > cache_t) return P3V_ERR_ARG;
  if (qkv && (!cos_t || !sin_t || n_heads <= 0 || n_kv <= 0 || BH % n_kv || tab_div <= 0)) return P3V_ERR_ARG;
  const Mlx4Src src = {qkv, cos_t, sin_t, n_tok, n_heads, n_kv, past, tab_t, tab_div};
  if (((uintptr_t)k | (uintptr_t)k4 | (uintptr_t)v4) & 15) return P3V_ERR_ARG;
  const long total = (long)BH * n_tok * (hd / 32) * 2;
  if (total == 0) return P3V_OK;
  hipLaunchKernelGGL(k_kv_quantize_mlx4, dim3((unsigned)p3v_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, k, vt, k4, v4, k_sb,
                     v_sb, hd, cache_t, n_tok, total, src);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- CLIP patch unfold + CLS rows
__global__ void k_im2col(const float* __restrict__ pix, bf16_t* __restrict__ out, int img, int patch, int grid, int kpad,
                         long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % kpad);
  const long r = i / kpad;
  const int pp = patch * patch;
  float v = 0.f;
  if (k < 3 * pp) {
    const int c = k / pp, ky = (k % pp) / patch, kx = k % patch;
    const int px = (int)(r % grid), py = (int)((r / grid) % grid);
    const long n = r / ((long)grid * grid);
    v = pix[((n * 3 + c) * img + (py * patch + ky)) * (long)img + px * patch + kx];
  }
  out[i] = f32_to_bf16(v);
}

extern "C" int p3v_im2col_patches(const float* pix, uint16_t* patches, int n_img, int img, int patch, int kpad,
                                  void* stream) {
  if (!pix || !patches || n_img < 0 || img % patch || kpad < 3 * patch * patch) return P3V_ERR_ARG;
  const int grid = img / patch;
  const long total = (long)n_img * grid * grid * kpad;
  if (total == 0) return P3V_OK;
  hipLaunchKernelGGL(k_im2col, dim3(p3v_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pix, patches, img, patch,
                     grid, kpad, total);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

__global__ void k_clip_cls(float* __restrict__ x, const bf16_t* __restrict__ cls, const bf16_t* __restrict__ pos,
                           int tokens, int dim) {
  const int n = blockIdx.x;
  for (int d = threadIdx.x; d < dim; d += blockDim.x)
    x[(size_t)n * tokens * dim + d] = bf16_to_f32(cls[d]) + bf16_to_f32(pos[d]);
}

extern "C" int p3v_clip_cls_rows(float* x, const uint16_t* cls, const uint16_t* pos, int n_img, int tokens, int dim,
                                 void* stream) {
  if (!x || !cls || !pos || n_img < 0) return P3V_ERR_ARG;
  if (n_img == 0) return P3V_OK;
  hipLaunchKernelGGL(k_clip_cls, dim3(n_img), dim3(256), 0, (hipStream_t)stream, x, cls, pos, tokens, dim);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- HD merge (pure index math)
// one block per output token; 4C channels = [dy][dx][C].
__global__ void __launch_bounds__(256) k_hd_merge(const float* __restrict__ feats, const bf16_t* __restrict__ sub_gn,
                                                  const bf16_t* __restrict__ glb_gn, bf16_t* __restrict__ out, int h,
                                                  int w, int grid, int C) {
  const int g2 = grid / 2;                       // 12
  const int sub_w = w * g2 + 1;
  const int n_sub = h * g2 * sub_w;
  const int t = blockIdx.x;
  bf16_t* o = out + (size_t)t * 4 * C;
  const bf16_t* gn = nullptr;
  int crop = 0, i = 0, j = 0;
  if (t < n_sub) {
    const int line = t / sub_w, col = t % sub_w;
    if (col == w * g2) gn = sub_gn;
    else {
      const int q = line * (w * g2) + col;       // crop-major token index (reference quirk Q4)
      crop = 1 + q / (g2 * g2);
      i = (q % (g2 * g2)) / g2;
      j = q % g2;
    }
  } else if (t == n_sub) {
    gn = glb_gn;
  } else {
    const int u = t - n_sub - 1, line = u / (g2 + 1), col = u % (g2 + 1);
    if (col == g2) gn = sub_gn;
    else { crop = 0; i = line; j = col; }
  }
  if (gn) {
    for (int c = threadIdx.x; c < 4 * C; c += blockDim.x) o[c] = gn[c];
    return;
  }
  const size_t tok_stride = C, crop_stride = (size_t)(grid * grid + 1) * C;
  for (int c4 = threadIdx.x; c4 < C; c4 += blockDim.x) {   // c4 indexes 4-wide f32 chunks over [dy][dx][C]
    const int e = c4 * 4, dy = e / (2 * C), dx = (e / C) & 1, c = e % C;
    const int p = (2 * i + dy) * grid + (2 * j + dx);
    const float4 v = *(const float4*)(feats + crop * crop_stride + (size_t)(p + 1) * tok_stride + c);
    u32x2_t r;
    r[0] = pack_bf16x2(v.x, v.y);
    r[1] = pack_bf16x2(v.z, v.w);
    *(u32x2_t*)(o + e) = r;
  }
}

extern "C" int p3v_hd_merge(const float* feats, const uint16_t* sub_gn, const uint16_t* glb_gn, uint16_t* out, int h_crops,
                            int w_crops, int grid, int C, void* stream) {
  if (!feats || !sub_gn || !glb_gn || !out || h_crops <= 0 || w_crops <= 0 || grid % 2 || C % 4) return P3V_ERR_ARG;
  const int g2 = grid / 2;
  const int n_out = h_crops * g2 * (w_crops * g2 + 1) + 1 + g2 * (g2 + 1);
  hipLaunchKernelGGL(k_hd_merge, dim3(n_out), dim3(256), 0, (hipStream_t)stream, feats, sub_gn, glb_gn, out, h_crops,
                     w_crops, grid, C);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- argmax / log-softmax / top-k over the vocab
struct ValIdx { float v; int i; };
__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b) {   // larger value, then smaller index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ValIdx block_argmax(ValIdx m, ValIdx* red) {
  // wave all-reduce of (value, index) on DPP + row swaps (see wave_sum in p3v_common.h)
#define P3V_ARGMAX_DPP(ctrl)                                                                             \
  {                                                                                                      \
    ValIdx t;                                                                                            \
    t.v = P3V_DPP_F32(m.v, ctrl);                                                                        \
    t.i = __builtin_amdgcn_update_dpp(0, m.i, ctrl, 0xf, 0xf, true);                                     \
    m = better(m, t);                                                                                    \
  }
  P3V_ARGMAX_DPP(0xB1) P3V_ARGMAX_DPP(0x4E) P3V_ARGMAX_DPP(0x124) P3V_ARGMAX_DPP(0x128)
#undef P3V_ARGMAX_DPP
  {
    ValIdx a, b;
    float ia, ib;
    rows_swap32(m.v, a.v, b.v);
    rows_swap32(__builtin_bit_cast(float, m.i), ia, ib);
    a.i = __builtin_bit_cast(int, ia); b.i = __builtin_bit_cast(int, ib);
    m = better(a, b);
    rows_swap16(m.v, a.v, b.v);
    rows_swap16(__builtin_bit_cast(float, m.i), ia, ib);
    a.i = __builtin_bit_cast(int, ia); b.i = __builtin_bit_cast(int, ib);
    m = better(a, b);
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = m;
  __syncthreads();
  ValIdx r = red[0];
  for (int k = 1; k < nw; ++k) r = better(r, red[k]);
  return r;
}

// per-thread argmax over a bf16 row with 16-byte loads (all of a thread's loads are independent: one memory round
// trip for a 32064-wide row on 1024 threads); first maximum wins
__device__ __forceinline__ ValIdx row_argmax_partial(const bf16_t* __restrict__ r, int n) {
  ValIdx m = {-INFINITY, 0x7fffffff};
  // A NaN logit means a kernel upstream failed (the split-KV merge poisons its output when its bounded wait runs out).
  // It enters the reduction as (+inf, index -1), which beats every real entry: the row's arg-max is then -1 instead of
  // an arbitrary index, and the host loops raise on a negative token (api._rows) -- loud, not a silently wrong token.
  auto take = [&](float v, int i) {
    if (v != v) { v = INFINITY; i = -1; }
    if (v > m.v || (v == m.v && i < m.i) || m.i == 0x7fffffff) { m.v = v; m.i = i; }
  };
  if ((((size_t)r) & 15) == 0) {
    const int nv = n >> 3;
    const u32x4_t* rv = (const u32x4_t*)r;
#pragma unroll 4
    for (int c = threadIdx.x; c < nv; c += blockDim.x) {
      const u32x4_t w = rv[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) { take(bf16lo(w[j]), 8 * c + 2 * j); take(bf16hi(w[j]), 8 * c + 2 * j + 1); }
    }
    for (int i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) take(bf16_to_f32(r[i]), i);
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) take(bf16_to_f32(r[i]), i);
  }
  return m;
}

__global__ void __launch_bounds__(1024) k_argmax(const bf16_t* __restrict__ x, int32_t* __restrict__ out, int n,
                                                 int64_t stride) {
  __shared__ ValIdx red[16];
  ValIdx m = row_argmax_partial(x + (size_t)blockIdx.x * stride, n);
  m = block_argmax(m, red);
  if (threadIdx.x == 0) out[blockIdx.x] = m.i;
}

extern "C" int p3v_argmax(const uint16_t* logits, int32_t* out, int rows, int n, int64_t row_stride, void* stream) {
  if (!logits || !out || rows < 0 || n <= 0) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  hipLaunchKernelGGL(k_argmax, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, out, n, row_stride);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

__global__ void __launch_bounds__(1024) k_log_softmax(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int n) {
  __shared__ float red[16];
  const bf16_t* r = x + (size_t)blockIdx.x * n;
  bf16_t* o = y + (size_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, bf16_to_f32(r[i]));
  m = block_max(m, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += expf(bf16_to_f32(r[i]) - m);
  s = block_sum(s, red);
  // nn.log_softmax(x) = x - mx.logsumexp(x): the normaliser is itself a bf16 array before the subtraction rounds again
  const float lse = bf16_round(m + logf(s));
  for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = f32_to_bf16(bf16_to_f32(r[i]) - lse);
}

extern "C" int p3v_log_softmax(const uint16_t* x, uint16_t* y, int rows, int n, void* stream) {
  if (!x || !y || rows < 0 || n <= 0) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  hipLaunchKernelGGL(k_log_softmax, dim3(rows), dim3(1024), 0, (hipStream_t)stream, x, y, n);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

__global__ void __launch_bounds__(1024) k_topk(const bf16_t* __restrict__ x, int32_t* __restrict__ out, int n, int k,
                                               int64_t stride) {
  __shared__ ValIdx red[16];
  __shared__ int picked[8];
  const bf16_t* r = x + (size_t)blockIdx.x * stride;
  for (int round = 0; round < k; ++round) {
    ValIdx m = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      bool skip = false;
      for (int p = 0; p < round; ++p) skip |= (picked[p] == i);
      if (skip) continue;
      const float v = bf16_to_f32(r[i]);
      if (v > m.v || (v == m.v && i < m.i) || m.i == 0x7fffffff) { m.v = v; m.i = i; }
    }
    m = block_argmax(m, red);
    if (threadIdx.x == 0) { picked[round] = m.i; out[blockIdx.x * k + round] = m.i; }
    __syncthreads();
  }
}

extern "C" int p3v_topk(const uint16_t* x, int32_t* idx_out, int rows, int n, int k, int64_t row_stride, void* stream) {
  if (!x || !idx_out || rows < 0 || n <= 0 || k <= 0 || k > 8 || k > n) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  hipLaunchKernelGGL(k_topk, dim3(rows), dim3(1024), 0, (hipStream_t)stream, x, idx_out, n, k, row_stride);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- device-resident loop state
__global__ void k_add_i32(int32_t* x, int n, int delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += delta;
}
extern "C" int p3v_add_i32(int32_t* x, int n, int delta, void* stream) {
  if (!x || n <= 0) return P3V_ERR_ARG;
  hipLaunchKernelGGL(k_add_i32, dim3(p3v_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, x, n, delta);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

__global__ void k_store_token(const int32_t* tok, int32_t* hist, const int32_t* d_step, int32_t* tok_next, int B,
                              int max_steps) {
  const int b = threadIdx.x;
  const int s = *d_step;
  if (b >= B) return;
  const int t = tok[b];
  if (s < max_steps) hist[(size_t)b * max_steps + s] = t;
  if (tok_next) tok_next[b] = t;
}
extern "C" int p3v_store_token(const int32_t* tok, int32_t* history, const int32_t* d_step, int32_t* tok_next, int B,
                               int max_steps, void* stream) {
  if (!tok || !history || !d_step || B <= 0 || B > 1024) return P3V_ERR_ARG;
  hipLaunchKernelGGL(k_store_token, dim3(1), dim3(p3v_cdiv(B, 64) * 64), 0, (hipStream_t)stream, tok, history, d_step,
                     tok_next, B, max_steps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- fused head / tail of a graph-replayed greedy step
// p3v_step_begin = embedding gather of the B current tokens + staging of the rotation-table rows of position
// *d_past (one workgroup per batch row); p3v_step_end = argmax + history/tok bookkeeping + the two counters.
// Four launches fewer per decode step than the separate entry points (a launch boundary costs ~2.4 us in-graph).
__global__ void __launch_bounds__(128) k_step_begin(const int32_t* __restrict__ ids, const u32x4_t* __restrict__ table,
                                                    u32x4_t* __restrict__ out, int chunks, int vocab,
                                                    const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                    const int32_t* __restrict__ d_past, float* __restrict__ cos_o,
                                                    float* __restrict__ sin_o, int tab_t, int half,
                                                    int32_t* __restrict__ zero_buf, int n_zero, int n_rows) {
  const int b = blockIdx.x;
  for (int i = b * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) zero_buf[i] = 0;
  if (b >= n_rows) return;                   // extra workgroups only help clearing zero_buf
  const int past = *d_past;
  int id = ids[b];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4_t* src = table + (size_t)id * chunks;
  u32x4_t* dst = out + (size_t)b * chunks;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) dst[c] = src[c];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    cos_o[(size_t)b * half + i] = cos_t[((size_t)b * tab_t + past) * half + i];
    sin_o[(size_t)b * half + i] = sin_t[((size_t)b * tab_t + past) * half + i];
  }
}

extern "C" int p3v_step_begin(const int32_t* tok, const uint16_t* table, uint16_t* x_out, const float* cos_t,
                              const float* sin_t, const int32_t* d_past, float* cos_out, float* sin_out, int B, int hidden,
                              int vocab, int tab_t, int half_dim, int32_t* zero_buf, int n_zero, void* stream) {
  if (!tok || !table || !x_out || !cos_t || !sin_t || !d_past || !cos_out || !sin_out) return P3V_ERR_ARG;
  if (B <= 0 || hidden % 8 || vocab <= 0 || tab_t <= 0 || half_dim <= 0) return P3V_ERR_ARG;
  const int n_wg = zero_buf && n_zero > 4096 ? (B > 128 ? B : 128) : B;
  hipLaunchKernelGGL(k_step_begin, dim3(n_wg), dim3(128), 0, (hipStream_t)stream, tok, (const u32x4_t*)table, (u32x4_t*)x_out,
                     hidden / 8, vocab, cos_t, sin_t, d_past, cos_out, sin_out, tab_t, half_dim, zero_buf, zero_buf ? n_zero : 0, B);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

__global__ void __launch_bounds__(1024) k_step_end(const bf16_t* __restrict__ logits, int32_t* __restrict__ next_tok,
                                                   int32_t* __restrict__ tok, int32_t* __restrict__ hist, int32_t* d_step,
                                                   int32_t* d_past, int32_t* ticket, int n, int max_steps) {
  __shared__ ValIdx red[16];
  const int b = blockIdx.x;
  const int s = *d_step;                     // read before this workgroup takes its ticket (the last one bumps it)
  const int past_now = *d_past;              // likewise (its round trip overlaps the scan instead of following it)
  ValIdx m = row_argmax_partial(logits + (size_t)b * n, n);
  m = block_argmax(m, red);
  if (threadIdx.x == 0) {
    next_tok[b] = m.i;
    tok[b] = m.i;
    if (s < max_steps) hist[(size_t)b * max_steps + s] = m.i;
    // The counters may only move once every workgroup has READ *d_step: the read above has returned (its value was
    // used), so a relaxed ticket is enough -- no cache write-back / invalidate (an agent-scope release costs ~20 us).
    bool last = gridDim.x == 1;
    if (!last) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    if (last) {
      *d_step = s + 1;
      *d_past = past_now + 1;
      if (gridDim.x > 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" int p3v_step_end(const uint16_t* logits, int32_t* next_tok, int32_t* tok, int32_t* history, int32_t* d_step,
                            int32_t* d_past, int32_t* ticket, int B, int n, int max_steps, void* stream) {
  if (!logits || !next_tok || !tok || !history || !d_step || !d_past || !ticket || B <= 0 || n <= 0) return P3V_ERR_ARG;
  hipLaunchKernelGGL(k_step_end, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, next_tok, tok, history, d_step, d_past,
                     ticket, n, max_steps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
