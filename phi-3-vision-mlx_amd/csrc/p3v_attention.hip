// Attention for the decoder (causal + left-pad, growing KV cache, head_dim 96;
// reference phi.py:454-457 with Mask4D phi.py:550-563) and for CLIP (no mask,
// head_dim 64; phi.py:148).  Scores and the mask are never materialised.
//
// One kernel, "swapped" MFMA formulation (per wave: 16 queries x 64-key tiles):
//   S^T[key,q] = K_tile[key,:] . Q[q,:]      v_mfma_f32_16x16x32_bf16, A = K rows from LDS, B = Q in VGPRs
//   online softmax: every lane owns ONE query column (q = lane&15) and 16 keys of
//     the tile, so max/sum need 15 in-lane ops + 2 shuffles and the rescale
//     factor is lane-local
//   O^T[d,q] += V^T[d,key] . P^T[key,q]      B = P straight from the S^T accumulators:
//     the MFMA k-index is permuted (k=(g,j) <-> key 32*st + 16*(j/4) + 4*g + j%4)
//     so that the P values a lane already holds ARE its B fragment; the same
//     permutation is applied when V^T is read from LDS.  P never leaves registers.
//
// Two launch shapes:
//   prefill  grid (ceil(L/64), heads, B): 4 waves = 4 query tiles sharing K/V tiles in LDS
//   decode   L <= 16: grid (n_split, heads, B): the KV range is split across
//            blocks (HBM-bound: reads 2*T*hd*2 bytes per head once); partial
//            (m, l, O) go to a workspace and `k_attn_combine` merges them.
#include "p3v_common.h"

struct AttnP {
  const bf16_t* q; const bf16_t* k_past; const bf16_t* v_past; const bf16_t* k_new; const bf16_t* v_new;
  bf16_t* out; const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, nh, nkv, past, past_t, past_div, new_t, pad_div, causal, split_mode, n_split, new_is_cache;
  float scale;
};

template <int HD>
__global__ void __launch_bounds__(256) k_attn(AttnP p) {
  constexpr int KSTR = HD * 2 + 16;        // bytes per K row in LDS (padded: conflict-free b128 fragment reads)
  constexpr int VSTR = 64 * 2 + 8;         // bytes per V^T row in LDS
  constexpr int NKS = HD / 32;             // k-steps of QK^T
  constexpr int NDT = HD / 16;             // 16-wide d tiles of O^T
  constexpr int CPR = HD / 8;              // 16-byte chunks per K/V row
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * KSTR];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[HD * VSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, qi = lane & 15;
  const int b = blockIdx.z, head = blockIdx.y, kvh = head / (p.nh / p.nkv);
  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b / p.pad_div] : 0;

  int q0, kv_begin, kv_end;
  bool active;
  if (p.split_mode) {
    q0 = 0;
    const int chunk = ((total + p.n_split - 1) / p.n_split + 63) & ~63;
    kv_begin = blockIdx.x * chunk;
    kv_end = min(total, kv_begin + chunk);
    active = wave == 0;
  } else {
    q0 = blockIdx.x * 64 + wave * 16;
    kv_begin = 0;
    kv_end = p.causal ? min(total, past + blockIdx.x * 64 + 64) : total;
    active = q0 < p.L;
  }
  if (pad > kv_begin) kv_begin = pad & ~63;       // tiles entirely inside the left padding are skipped

  const int qrow = q0 + qi;
  const bool qvalid = qrow < p.L;
  bf16x8_t qf[NKS];
  {
    const bf16_t* qp = p.q + (((size_t)b * p.nh + head) * p.L + (qvalid ? qrow : 0)) * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = *(const u32x4_t*)(qp + 32 * ks);
      if (!qvalid) v = (u32x4_t){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const int qpos = past + qrow;                    // absolute position of this lane's query

  const bf16_t* kp_base = p.k_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)p.past_t * HD;
  const bf16_t* vp_base = p.v_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)p.past_t * HD;
  const bf16_t* kn_base = p.k_new + ((size_t)b * p.nkv + kvh) * (size_t)p.new_t * HD;
  const bf16_t* vn_base = p.v_new + ((size_t)b * p.nkv + kvh) * (size_t)p.new_t * HD;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * CPR; i += 256) {
      const int key = i / CPR, c = i % CPR, t = kv0 + key;
      u32x4_t kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
      if (t < kv_end) {
        const bool from_past = t < past || p.new_is_cache;
        const bf16_t* ks = from_past ? kp_base + (size_t)t * HD : kn_base + (size_t)(t - past) * HD;
        const bf16_t* vs = from_past ? vp_base + (size_t)t * HD : vn_base + (size_t)(t - past) * HD;
        kv = *(const u32x4_t*)(ks + c * 8);
        vv = *(const u32x4_t*)(vs + c * 8);
      }
      *(u32x4_t*)(Ks + key * KSTR + c * 16) = kv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(bf16_t*)(Vt + (c * 8 + 2 * j) * VSTR + key * 2) = (bf16_t)(vv[j] & 0xffff);
        *(bf16_t*)(Vt + (c * 8 + 2 * j + 1) * VSTR + key * 2) = (bf16_t)(vv[j] >> 16);
      }
    }
    __syncthreads();
    if (!active) continue;

    // ---- S^T = K . Q^T for the 4 key sub-tiles
    f32x4_t s[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s[st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + (16 * st + qi) * KSTR + (32 * ks + 8 * g) * 2);
        s[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[st], 0, 0, 0);
      }
    }
    // ---- mask, running max
    float m_t = -INFINITY;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + 16 * st + 4 * g + r;
        const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos) && qpos >= pad;
        const float v = vis ? s[st][r] * p.scale : -INFINITY;
        s[st][r] = v;
        m_t = fmaxf(m_t, v);
      }
    m_t = fmaxf(m_t, __shfl_xor(m_t, 16, 64));
    m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
    const float m_new = fmaxf(m_run, m_t);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_use);       // m_run = -inf -> 0
    float l_t = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[st][r] - m_use);
        s[st][r] = e;
        l_t += e;
      }
    l_t += __shfl_xor(l_t, 16, 64);
    l_t += __shfl_xor(l_t, 32, 64);
    l_run = l_run * alpha + l_t;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[d] *= alpha;
    // ---- O^T += V^T . P^T  (two 32-key steps; see the k-index permutation in the header)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4_t pw;
      pw[0] = pack_bf16x2(s[2 * st][0], s[2 * st][1]);
      pw[1] = pack_bf16x2(s[2 * st][2], s[2 * st][3]);
      pw[2] = pack_bf16x2(s[2 * st + 1][0], s[2 * st + 1][1]);
      pw[3] = pack_bf16x2(s[2 * st + 1][2], s[2 * st + 1][3]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const unsigned char* vr = Vt + (16 * d + qi) * VSTR + (32 * st + 4 * g) * 2;
        const u32x2_t a0 = *(const u32x2_t*)vr, a1 = *(const u32x2_t*)(vr + 32);
        const u32x4_t aw = {a0[0], a0[1], a1[0], a1[1]};
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aw), pf, o[d], 0, 0, 0);
      }
    }
  }

  if (!active || !qvalid) return;
  if (p.split_mode) {
    float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + blockIdx.x) * 16 + qi) * (HD + 2);
#pragma unroll
    for (int d = 0; d < NDT; ++d) *(f32x4_t*)(w + 16 * d + 4 * g) = o[d];
    if (g == 0) { w[HD] = m_run; w[HD + 1] = l_run; }
  } else {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t* op = p.out + ((size_t)b * p.L + qrow) * (size_t)(p.nh * HD) + head * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      u32x2_t w;
      w[0] = pack_bf16x2(o[d][0] * inv, o[d][1] * inv);
      w[1] = pack_bf16x2(o[d][2] * inv, o[d][3] * inv);
      *(u32x2_t*)(op + 16 * d) = w;
    }
  }
}

// merge the split-KV partials: one block per (b, head, query)
__global__ void __launch_bounds__(128) k_attn_combine(const float* __restrict__ ws, bf16_t* __restrict__ out, int L, int nh,
                                                      int hd, int n_split) {
  const int qi = blockIdx.x % L, head = (blockIdx.x / L) % nh, b = blockIdx.x / (L * nh);
  const float* base = ws + (((size_t)b * nh + head) * n_split * 16 + qi) * (hd + 2);
  const size_t sstr = (size_t)16 * (hd + 2);
  float M = -INFINITY;
  for (int s = 0; s < n_split; ++s) M = fmaxf(M, base[s * sstr + hd]);
  const int d = threadIdx.x;
  if (d >= hd) return;
  float acc = 0.f, l = 0.f;
  if (M > -INFINITY) {
    for (int s = 0; s < n_split; ++s) {
      const float m = base[s * sstr + hd];
      if (m == -INFINITY) continue;
      const float wgt = __expf(m - M);
      acc += wgt * base[s * sstr + d];
      l += wgt * base[s * sstr + hd + 1];
    }
  }
  out[((size_t)b * L + qi) * (size_t)(nh * hd) + head * hd + d] = f32_to_bf16(l > 0.f ? acc / l : 0.f);
}

extern "C" int64_t p3v_attention_ws_bytes(int B, int L, int n_heads, int hd, int n_split) {
  if (L > P3V_DECODE_MAX_L || n_split <= 1) return 0;
  return (int64_t)B * n_heads * n_split * 16 * (hd + 2) * 4;
}

extern "C" int p3v_attention(const p3v_attn_args_t* a, void* stream) {
  if (!a || !a->q || !a->out) return P3V_ERR_ARG;
  if (a->new_is_cache ? (!a->k_past || !a->v_past) : (!a->k_new || !a->v_new)) return P3V_ERR_ARG;
  if (a->hd != 64 && a->hd != 96) return P3V_ERR_UNSUPPORTED;
  if (a->B < 0 || a->L < 0 || a->n_heads <= 0 || a->n_kv <= 0 || a->n_heads % a->n_kv) return P3V_ERR_ARG;
  if ((a->past > 0 || a->d_past) && (!a->k_past || !a->v_past)) return P3V_ERR_ARG;
  if (a->B * a->L == 0) return P3V_OK;
  AttnP p;
  p.q = a->q; p.k_past = a->k_past ? a->k_past : a->k_new; p.v_past = a->v_past ? a->v_past : a->v_new;
  p.k_new = a->new_is_cache ? a->k_past : a->k_new; p.v_new = a->new_is_cache ? a->v_past : a->v_new; p.new_is_cache = a->new_is_cache;
  p.out = a->out; p.pad_len = a->pad_len; p.d_past = a->d_past; p.ws = a->ws;
  p.B = a->B; p.L = a->L; p.nh = a->n_heads; p.nkv = a->n_kv; p.past = a->past; p.past_t = a->past_t;
  p.past_div = a->past_div > 0 ? a->past_div : 1; p.new_t = a->new_t; p.pad_div = a->pad_div > 0 ? a->pad_div : 1;
  p.causal = a->causal; p.scale = a->scale;
  p.n_split = a->n_split;
  p.split_mode = (a->L <= P3V_DECODE_MAX_L && a->n_split > 1) ? 1 : 0;
  if (p.split_mode && !a->ws) return P3V_ERR_ARG;
  if (!p.split_mode) p.n_split = 1;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.split_mode ? p.n_split : p3v_cdiv(a->L, 64), a->n_heads, a->B);
  if (a->hd == 96) hipLaunchKernelGGL(k_attn<96>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_attn<64>, grid, dim3(256), 0, s, p);
  P3V_CHECK_LAUNCH();
  if (p.split_mode) {
    hipLaunchKernelGGL(k_attn_combine, dim3(a->B * a->n_heads * a->L), dim3(128), 0, s, a->ws, a->out, a->L, a->n_heads,
                       a->hd, p.n_split);
    P3V_CHECK_LAUNCH();
  }
  return P3V_OK;
}
