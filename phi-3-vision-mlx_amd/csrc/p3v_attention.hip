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
#include <stdlib.h>

#include "p3v_common.h"
#include "p3v_gemv3_body.h"      // dot8: the fused o_proj of k_attn_decode128_o repeats k_gemv3's arithmetic exactly
#include "p3v_dot_f8.h"          // dot16_f8: ... k_attn_decode128_q8<true> repeats k_gemv3_f8's
#include "p3v_dot_q4.h"          // dot8_q4: ... and k_attn_decode128_o4 repeats k_gemv3_q4's

struct AttnP {
  const bf16_t* q; const bf16_t* k_past; const bf16_t* v_past; const bf16_t* k_new; const bf16_t* v_new;
  bf16_t* out; const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, nh, nkv, past, past_t, past_div, new_t, pad_div, causal, split_mode, n_split, new_is_cache;
  float scale;
  int head_group;              // k_attn_prefill_dma / _pp: heads whose query blocks are interleaved in launch order
  float sc2;                   // what S = Q.K^T is multiplied by before exp2: scale * log2(e), or 1 when Q arrives pre-scaled
  int q_prescaled;
};

template <int HD>
__global__ void __launch_bounds__(256) k_attn(AttnP p) {
  constexpr int KSTR = HD * 2 + 16;        // bytes per K row in LDS (padded: conflict-free b128 fragment reads)
  constexpr int VSTR = 64 * 2 + 16;        // bytes per V^T row in LDS (16-B aligned rows, conflict-free b64 reads)
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
  const float sc2 = p.sc2;                                     // scale * log2(e): softmax (and the split partials) run on exp2

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
  const bf16_t* vp_base = p.v_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)HD * p.past_t;   // V^T: [hd][past_t]
  const bf16_t* kn_base = p.k_new + ((size_t)b * p.nkv + kvh) * (size_t)p.new_t * HD;
  const bf16_t* vn_base = p.v_new + ((size_t)b * p.nkv + kvh) * (size_t)HD * p.new_t;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * CPR; i += 256) {               // K tile: 64 key rows x CPR 16-byte chunks
      const int key = i / CPR, c = i % CPR, t = kv0 + key;
      u32x4_t kv = {0, 0, 0, 0};
      if (t < kv_end) {
        const bool from_past = t < past || p.new_is_cache;
        const bf16_t* ks = from_past ? kp_base + (size_t)t * HD : kn_base + (size_t)(t - past) * HD;
        kv = *(const u32x4_t*)(ks + c * 8);
      }
      *(u32x4_t*)(Ks + key * KSTR + c * 16) = kv;
    }
    for (int i = tid; i < HD * 8; i += 256) {                 // V^T tile: HD rows x 8 chunks of 8 keys
      const int d = i >> 3, c = i & 7, t0 = kv0 + c * 8;
      u32x4_t vv = {0, 0, 0, 0};
      if (t0 < kv_end) {
        if (t0 + 8 <= past || p.new_is_cache) {
          vv = *(const u32x4_t*)(vp_base + (size_t)d * p.past_t + t0);
        } else {                                              // chunk touches the separate "new" segment (beam path)
          bf16_t e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            e[j] = t < past ? vp_base[(size_t)d * p.past_t + t] : (t < kv_end ? vn_base[(size_t)d * p.new_t + (t - past)] : (bf16_t)0);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) vv[j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
        }
      }
      *(u32x4_t*)(Vt + d * VSTR + c * 16) = vv;
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
        const float v = vis ? s[st][r] * sc2 : -INFINITY;                 // log2 domain
        s[st][r] = v;
        m_t = fmaxf(m_t, v);
      }
    m_t = rows_max(m_t);
    const float m_new = fmaxf(m_run, m_t);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);       // m_run = -inf -> 0
    float l_t = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[st][r] - m_use);
        s[st][r] = e;
        l_t += e;
      }
    l_t = rows_sum(l_t);
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

constexpr int CMB_G = 16;   // max 64-lane groups of the split-KV merge kernel
static inline int combine_threads(int n_split) {              // measured at 41 splits: 8 groups beat 4 and 16
  const int g = p3v_tuning().combine_g;                       // tuning knob (4 / 8 / 16)
  if (g > 0) return 64 * g;
  return 64 * (n_split <= 8 ? 4 : 8);
}
__global__ void k_attn_combine2(const float* __restrict__ ws, bf16_t* __restrict__ out, int L, int nh, int hd, int n_split);
// =====================================================================================
// Prefill / CLIP attention (L > 16): 128 queries per workgroup (4 waves x 2 sub-tiles of 16), 64-key tiles,
// two LDS buffers.  The next tile's global loads are issued BEFORE the MFMAs of the current tile and written
// to the other LDS buffer after them (split stage: HBM/L2 latency hides under the matrix work), one barrier
// per tile.  K and V^T fragments read from LDS are shared by the wave's two query sub-tiles (48 MFMAs per
// 24 + 24 fragment reads).  Same swapped-MFMA formulation and masks as k_attn above.
template <int HD>
__global__ void __launch_bounds__(256, 2) k_attn_prefill(AttnP p) {
  constexpr int KSTR = HD * 2 + 16, VSTR = 64 * 2 + 16, NKS = HD / 32, NDT = HD / 16, CPR = HD / 8;
  constexpr int KTILE = 64 * KSTR, VTILE = HD * VSTR, BUF = KTILE + VTILE;
  constexpr int NLD = (64 * CPR + 255) / 256;               // 16-byte K loads per thread per tile (= V^T loads)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, qi = lane & 15;
  const int b = blockIdx.z, head = blockIdx.y, kvh = head / (p.nh / p.nkv);
  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b / p.pad_div] : 0;
  const int qb0 = blockIdx.x * 128, q0 = qb0 + wave * 32;
  const int kv_end = p.causal ? min(total, past + qb0 + 128) : total;
  const int kv_begin = pad & ~63;
  const int wave_last = p.causal ? past + q0 + 31 : total;  // last key position this wave can see
  const float sc2 = p.sc2;                                            // scale * log2(e)

  bf16x8_t qf[2][NKS];
  int qpos[2];
  bool qvalid[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = q0 + u * 16 + qi;
    qvalid[u] = qrow < p.L;
    qpos[u] = past + qrow;
    const bf16_t* qp = p.q + (((size_t)b * p.nh + head) * p.L + (qvalid[u] ? qrow : 0)) * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = *(const u32x4_t*)(qp + 32 * ks);
      if (!qvalid[u]) v = (u32x4_t){0, 0, 0, 0};
      qf[u][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const bf16_t* kp_base = p.k_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)p.past_t * HD;
  const bf16_t* vp_base = p.v_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)HD * p.past_t;
  const bf16_t* kn_base = p.k_new + ((size_t)b * p.nkv + kvh) * (size_t)p.new_t * HD;
  const bf16_t* vn_base = p.v_new + ((size_t)b * p.nkv + kvh) * (size_t)HD * p.new_t;

  u32x4_t kst[NLD], vst[NLD];
  auto stage_load = [&](int kv0) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = it * 256 + tid;
      kst[it] = (u32x4_t){0, 0, 0, 0};
      vst[it] = (u32x4_t){0, 0, 0, 0};
      if (i < 64 * CPR) {
        const int key = i / CPR, c = i % CPR, t = kv0 + key;
        if (t < kv_end) {
          const bool fp = t < past || p.new_is_cache;
          kst[it] = *(const u32x4_t*)((fp ? kp_base + (size_t)t * HD : kn_base + (size_t)(t - past) * HD) + c * 8);
        }
        const int d = i >> 3, c8 = i & 7, t0 = kv0 + c8 * 8;       // HD*8 == 64*CPR chunks as well
        if (t0 < kv_end) {
          if (t0 + 8 <= past || p.new_is_cache) {
            vst[it] = *(const u32x4_t*)(vp_base + (size_t)d * p.past_t + t0);
          } else {
            bf16_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int t = t0 + j;
              e[j] = t < past ? vp_base[(size_t)d * p.past_t + t] : (t < kv_end ? vn_base[(size_t)d * p.new_t + (t - past)] : (bf16_t)0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) vst[it][j] = (uint32_t)e[2 * j] | ((uint32_t)e[2 * j + 1] << 16);
          }
        }
      }
    }
  };
  auto stage_write = [&](int buf) {
    unsigned char* Ks = smem + buf * BUF;
    unsigned char* Vt = Ks + KTILE;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = it * 256 + tid;
      if (i < 64 * CPR) {
        *(u32x4_t*)(Ks + (i / CPR) * KSTR + (i % CPR) * 16) = kst[it];
        *(u32x4_t*)(Vt + (i >> 3) * VSTR + (i & 7) * 16) = vst[it];
      }
    }
  };

  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  f32x4_t o[2][NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[u][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kv_begin < kv_end) { stage_load(kv_begin); stage_write(0); }
  __syncthreads();
  int buf = 0;
  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += 64, buf ^= 1) {
    const bool more = kv0 + 64 < kv_end;
    if (more) stage_load(kv0 + 64);                         // in flight during the MFMAs below
    if (kv0 <= wave_last) {                                 // wave-uniform: tiles above this wave's diagonal are skipped
      const unsigned char* Ks = smem + buf * BUF;
      const unsigned char* Vt = Ks + KTILE;
      f32x4_t s[2][4];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s[0][st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        s[1][st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8_t kf = *(const bf16x8_t*)(Ks + (16 * st + qi) * KSTR + (32 * ks + 8 * g) * 2);
          s[0][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[0][st], 0, 0, 0);
          s[1][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[1][st], 0, 0, 0);
        }
      }
      // every key of the tile visible to every query of the wave -> no per-element mask work (wave-uniform)
      const bool interior = kv0 + 64 <= kv_end && kv0 >= pad && past + q0 >= pad && (!p.causal || kv0 + 63 <= past + q0);
      bf16x8_t pf[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float m_t = -INFINITY;
        if (interior) {
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) m_t = fmaxf(m_t, s[u][st][r]);
        } else {
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int t = kv0 + 16 * st + 4 * g + r;
              const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos[u]) && qpos[u] >= pad;
              const float v = vis ? s[u][st][r] : -INFINITY;
              s[u][st][r] = v;
              m_t = fmaxf(m_t, v);
            }
        }
        m_t = rows_max(m_t);
        // running max / exponentials in the log2 domain: exp(x*scale - m) = exp2(x*c - m2), c = scale*log2(e)
        const float m_new = fmaxf(m_run[u], m_t * sc2);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_use);
        float l_t = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[u][st][r], sc2, -m_use));
            s[u][st][r] = e;
            l_t += e;
          }
        l_t = rows_sum(l_t);
        l_run[u] = l_run[u] * alpha + l_t;
        m_run[u] = m_new;
        if (!__all(alpha == 1.f)) {
#pragma unroll
          for (int d = 0; d < NDT; ++d) o[u][d] *= alpha;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          u32x4_t pw;
          pw[0] = pack_bf16x2(s[u][2 * st][0], s[u][2 * st][1]);
          pw[1] = pack_bf16x2(s[u][2 * st][2], s[u][2 * st][3]);
          pw[2] = pack_bf16x2(s[u][2 * st + 1][0], s[u][2 * st + 1][1]);
          pw[3] = pack_bf16x2(s[u][2 * st + 1][2], s[u][2 * st + 1][3]);
          pf[u][st] = __builtin_bit_cast(bf16x8_t, pw);
        }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const unsigned char* vr = Vt + (16 * d + qi) * VSTR + (32 * st + 4 * g) * 2;
          const u32x2_t a0 = *(const u32x2_t*)vr, a1 = *(const u32x2_t*)(vr + 32);
          const u32x4_t aw = {a0[0], a0[1], a1[0], a1[1]};
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, aw);
          o[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][st], o[0][d], 0, 0, 0);
          o[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][st], o[1][d], 0, 0, 0);
        }
    }
    if (more) stage_write(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!qvalid[u]) continue;
    const float inv = l_run[u] > 0.f ? 1.f / l_run[u] : 0.f;
    bf16_t* op = p.out + ((size_t)b * p.L + (q0 + u * 16 + qi)) * (size_t)(p.nh * HD) + head * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      u32x2_t w;
      w[0] = pack_bf16x2(o[u][d][0] * inv, o[u][d][1] * inv);
      w[1] = pack_bf16x2(o[u][d][2] * inv, o[u][d][3] * inv);
      *(u32x2_t*)(op + 16 * d) = w;
    }
  }
}

// Prefill / CLIP attention over a cache whose capacity is a multiple of 64 (every caller in the model): the K tile
// (64 rows x 2*HD bytes, contiguous) and the V^T tile (HD rows x 128 bytes) go HBM/L2 -> LDS by LDS-DMA, each wave
// issuing a quarter of the tile's 1 KiB pieces -- no staging registers, no ds_write pass, no per-tile address
// arithmetic (k_attn_prefill spends most of its issue slots there).  Bank-conflict swizzles on the source side:
//   192-byte K rows (HD = 96): chunk c ^ (-(row >> 2) & 3) (round 3: the round-2 form (row >> 2) & 3 was 2-way conflicted on ds_read_b128);   128-byte rows (K at HD = 64, V^T): chunk c ^ ((row >> 1) & 7).
// Two LDS buffers, one barrier per tile; compute part and masks identical to k_attn_prefill.
typedef const __attribute__((address_space(1))) void* pf_gptr_t;
typedef __attribute__((address_space(3))) void* pf_lptr_t;
template <int HD, bool PRE>                                    // PRE: Q arrives multiplied by scale * log2(e) (p3v_rope_kv_append's q_scale)
__global__ void __launch_bounds__(256, 2) k_attn_prefill_dma(AttnP p) {
  constexpr int KROW = HD * 2, VROW = 128, NKS = HD / 32, NDT = HD / 16, CPR = HD / 8;
  constexpr int KTILE = 64 * KROW, VTILE = HD * VROW, BUF = KTILE + VTILE;
  constexpr int NK = KTILE / 1024, NV = VTILE / 1024;         // 1 KiB DMA pieces per tile (K, V^T); NK % 4 == NV % 4 == 0
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Launch order (1-D over heads x query blocks): query blocks are walked from the LAST one down -- under a causal mask
  // block i costs i+1 tiles, so the dispatcher sees the longest jobs first and the tail of the launch is made of short
  // ones -- interleaving `head_group` heads at a time, so that the K/V of the heads in flight stay cache-resident
  // (all heads for short prompts; 8 or 4 when one head's K/V is megabytes).
  const int nqb = (p.L + 127) >> 7, per_group = p.head_group * nqb;
  const int grp = blockIdx.x / per_group, within = blockIdx.x - grp * per_group;
  const int qblk = nqb - 1 - within / p.head_group;
  const int b = blockIdx.z, head = grp * p.head_group + within % p.head_group, kvh = head / (p.nh / p.nkv);
  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b / p.pad_div] : 0;
  const int qb0 = qblk * 128, q0 = qb0 + wave * 32;
  const int kv_end = p.causal ? min(total, past + qb0 + 128) : total;
  const int kv_begin = pad & ~63;
  const int wave_last = p.causal ? past + q0 + 31 : total;  // last key position this wave can see
  const float sc2 = p.sc2;                                            // scale * log2(e)

  bf16x8_t qf[2][NKS];
  int qpos[2];
  bool qvalid[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = q0 + u * 16 + qi;
    qvalid[u] = qrow < p.L;
    qpos[u] = past + qrow;
    const bf16_t* qp = p.q + (((size_t)b * p.nh + head) * p.L + (qvalid[u] ? qrow : 0)) * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = *(const u32x4_t*)(qp + 32 * ks);
      if (!qvalid[u]) v = (u32x4_t){0, 0, 0, 0};
      qf[u][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const unsigned char* kbase = (const unsigned char*)(p.k_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)p.past_t * HD);
  const unsigned char* vbase = (const unsigned char*)(p.v_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)HD * p.past_t);

  // ---- per-lane DMA source offsets of this wave's pieces (piece j = wave + 4*jj covers LDS slots 64j .. 64j+63)
  unsigned koff[NK / 4];
#pragma unroll
  for (int jj = 0; jj < NK / 4; ++jj) {
    const int i = (wave + 4 * jj) * 64 + lane, row = i / CPR, pc = i - row * CPR;
    const int sw = HD == 96 ? (-(row >> 2)) & 3 : (row >> 1) & 7;   // conflict-free on ds_read_b128's lane groups (see k_attn_prefill_pp)
    koff[jj] = row * KROW + ((pc ^ sw) << 4);
  }
  const unsigned vrow = (unsigned)p.past_t * 2;
  unsigned voff[NV / 4];
#pragma unroll
  for (int jj = 0; jj < NV / 4; ++jj) {
    const int i = (wave + 4 * jj) * 64 + lane, row = i >> 3, pc = i & 7;
    voff[jj] = (unsigned)row * vrow + ((pc ^ ((row >> 1) & 7)) << 4);
  }
  // MUBUF form of the LDS-DMA (a pending `global_load_lds` makes the compiler turn every later wait into vmcnt(0) /
  // lgkmcnt(0): the fragment reads of the tile in work could not be counted while the next tile was in flight)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0xffffffff, 0x00020000);
  auto stage = [&](int kv0, int buf) {
    unsigned char* Ks = smem + buf * BUF;
    // (every argument of the builtin is copied into a local of non-dependent type: with a type-dependent argument -- an
    // element of an array whose size depends on HD -- the call is only checked at instantiation, fails there in the HOST pass,
    // where the builtin does not exist, and hipcc silently drops the kernel's host stub)
    const int so_k = kv0 * KROW, so_v = kv0 * 2;
#pragma unroll
    for (int jj = 0; jj < NK / 4; ++jj) {
      const unsigned vo = koff[jj];
      pf_lptr_t dst = (pf_lptr_t)(Ks + (wave + 4 * jj) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dst, 16, vo, so_k, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < NV / 4; ++jj) {
      const unsigned vo = voff[jj];
      pf_lptr_t dst = (pf_lptr_t)(Ks + KTILE + (wave + 4 * jj) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dst, 16, vo, so_v, 0, 0);
    }
  };

  // ---- fragment read offsets
  unsigned k_rd[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int c = 4 * ks + g, sw = HD == 96 ? (-(qi >> 2)) & 3 : (qi >> 1) & 7;
    k_rd[ks] = qi * KROW + ((c ^ sw) << 4);                  // + st*16*KROW
  }
  unsigned v_rd[4];                                           // chunk (4 st + 2 h + g/2) ^ ((qi>>1)&7)
#pragma unroll
  for (int k = 0; k < 4; ++k) v_rd[k] = KTILE + qi * VROW + (((2 * k + (g >> 1)) ^ ((qi >> 1) & 7)) << 4) + (g & 1) * 8;   // + d*16*VROW

  // The row sums l ride on the matrix cores: one extra O^T tile whose V^T fragment is a register constant (row 0 all ones,
  // the other 15 rows zero), so l = P . 1 accumulates -- and is rescaled -- like any other output row, in lanes g = 0.
  // 4 MFMAs per tile instead of 32 v_add + two cross-lane reductions: the kernel is VALU-bound, the matrix pipe is 70 % idle.
  const u32x4_t ones_w = qi == 0 ? (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : (u32x4_t){0u, 0u, 0u, 0u};
  const bf16x8_t ones_f = __builtin_bit_cast(bf16x8_t, ones_w);
  float m_run[2] = {PRE ? 0.f : -INFINITY, PRE ? 0.f : -INFINITY};   // PRE: reference of the exponent (finite always), log2 units; else: running maximum
  unsigned long long unset[2] = {~0ull, ~0ull};              // lanes whose query has not seen a visible key yet: the next one sets the reference
  f32x4_t negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // (PRE) -reference, the C operand of the S^T products
  f32x4_t ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4_t o[2][NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[u][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kv_begin < kv_end) stage(kv_begin, 0);
  int buf = 0;
  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += 64, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the tile have landed ...
    __syncthreads();                                          // ... everybody's have, and the other buffer is free
    if (kv0 + 64 < kv_end) stage(kv0 + 64, buf ^ 1);          // next tile streams in under the MFMAs below
    if (kv0 <= wave_last) {                                   // wave-uniform: tiles above this wave's diagonal are skipped
      const unsigned char* Ks = smem + buf * BUF;
      f32x4_t s[2][4];
      // the two MFMA blocks of a tile run at raised wave priority: the SIMD's other wave is usually in its softmax, and a
      // wave that has to wait for VALU slots between its MFMAs also loses the matrix pipe (8k: 650 -> 630 us)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s[0][st] = PRE ? negm[0] : (f32x4_t){0.f, 0.f, 0.f, 0.f};   // PRE: the accumulators START at -reference: the product IS the exponent
        s[1][st] = PRE ? negm[1] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8_t kf = *(const bf16x8_t*)(Ks + k_rd[ks] + st * 16 * KROW);
          s[0][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[0][st], 0, 0, 0);
          s[1][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[1][st], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      // every key of the tile visible to every query of the wave -> no per-element mask work (wave-uniform)
      const bool interior = kv0 + 64 <= kv_end && kv0 >= pad && past + q0 >= pad && (!p.causal || kv0 + 63 <= past + q0);
      bf16x8_t pf[2][2];
      // PRE (pre-scaled Q: the decoder's prompts): softmax with a DEFERRED reference, as k_attn_prefill_pp / _il -- the S^T
      // accumulators started at -reference, so the product IS the exponent; the reference moves only on the query's first
      // visible tile and when a score exceeds it by more than 2^8 (wave-uniform slow path: O, l and the tile's scores are
      // rescaled once, in place).  Fast path: 16 max + 32 v_exp + 16 v_cvt_pk against ~280 VALU of the classic form below
      // (row maximum across lanes, new maximum, rescale factor and 28 multiplies on EVERY tile).
      // Plain Q (the ViT, external callers) keeps the classic form: the deferred form was built for it too (CLIP 62 -> 55 us)
      // and withdrawn -- correct and bit-stable once its multiply-add was written in place (as `s = fmaf(s, sc2, -m)` the
      // compiler put the results into the registers the last S^T MFMA had read and the output became NON-DETERMINISTIC:
      // 16-query halves wrong by up to 0.8 in some launches, tools/attn_determinism.py), but a different rounding sequence in
      // the ViT moves the heavy-tailed fixtures' logit error by +-40 % (chaotic amplification, profiles/r03_heavy_tail.txt)
      // past tolerances calibrated on one implementation.
      if constexpr (PRE) {
      // (the first readers of the S^T accumulators below are INLINE-ASM instructions: hipcc pads the MFMA -> reader wait states
      //  -- 12 for this 8-pass MFMA, guide 5.7 item 2 -- only for instructions it emits itself.  Today ~70 instructions of mask
      //  / interior arithmetic sit in between; the explicit pad keeps that from being an accident of the code layout.)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7\n\ts_nop 4");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (!interior) {
          const float ninf = -INFINITY;
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int t = kv0 + 16 * st + 4 * g + r;
              const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos[u]) && qpos[u] >= pad;
              const unsigned long long vm = __builtin_amdgcn_ballot_w64(vis);
              asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(s[u][st][r]) : "v"(ninf), "s"(vm));      // s = vis ? s : -inf
            }
        }
        float ma, mc;                                          // this LANE's maximum over its 16 scores (no cross-lane step on the fast path)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(ma) : "v"(s[u][0][0]), "v"(s[u][0][1]), "v"(s[u][0][2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mc) : "v"(s[u][0][3]), "v"(s[u][1][0]), "v"(s[u][1][1]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(ma) : "v"(s[u][1][2]), "v"(s[u][1][3]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mc) : "v"(s[u][2][0]), "v"(s[u][2][1]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(ma) : "v"(s[u][2][2]), "v"(s[u][2][3]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mc) : "v"(s[u][3][0]), "v"(s[u][3][1]));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(ma) : "v"(s[u][3][2]), "v"(s[u][3][3]));
        asm("v_max_f32 %0, %0, %1" : "+v"(ma) : "v"(mc));
        if ((unset[u] | __builtin_amdgcn_ballot_w64(ma > 8.f)) != 0) {       // rare, wave-uniform: a reference moves
          const float m_t = rows_max(ma);
          const bool masked = m_t == -INFINITY;
          const bool un = (unset[u] >> lane) & 1;
          const float delta = un ? (masked ? 0.f : m_t) : fmaxf(m_t, 0.f);
          const float alpha = un ? 1.f : __builtin_amdgcn_exp2f(-delta);
          unset[u] &= __builtin_amdgcn_ballot_w64(masked);
          m_run[u] += delta;
          if (PRE) {
            const float nm = -m_run[u];                        // (in place: a new value would cost the fast path register copies)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm("v_mov_b32 %0, %1" : "+v"(negm[u][r]) : "v"(nm));
          }
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm("v_sub_f32 %0, %0, %1" : "+v"(s[u][st][r]) : "v"(delta));
#pragma unroll
          for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %0, %1" : "+v"(o[u][d][r]) : "v"(alpha));
#pragma unroll
          for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %0, %1" : "+v"(ol[u][r]) : "v"(alpha));
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][st][r] = __builtin_amdgcn_exp2f(s[u][st][r]);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          u32x4_t pw;
          pw[0] = pack_bf16x2(s[u][2 * st][0], s[u][2 * st][1]);
          pw[1] = pack_bf16x2(s[u][2 * st][2], s[u][2 * st][3]);
          pw[2] = pack_bf16x2(s[u][2 * st + 1][0], s[u][2 * st + 1][1]);
          pw[3] = pack_bf16x2(s[u][2 * st + 1][2], s[u][2 * st + 1][3]);
          pf[u][st] = __builtin_bit_cast(bf16x8_t, pw);
        }
      }
      } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float m_t = -INFINITY;
        if (interior) {
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) m_t = fmaxf(m_t, s[u][st][r]);
        } else {
#pragma unroll
          for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int t = kv0 + 16 * st + 4 * g + r;
              const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos[u]) && qpos[u] >= pad;
              const float v = vis ? s[u][st][r] : -INFINITY;
              s[u][st][r] = v;
              m_t = fmaxf(m_t, v);
            }
        }
        m_t = rows_max(m_t);
        const float m_new = fmaxf(m_run[u], m_t * sc2);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_use);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][st][r] = __builtin_amdgcn_exp2f(fmaf(s[u][st][r], sc2, -m_use));
        m_run[u] = m_new;
        if (!__all(alpha == 1.f)) {
#pragma unroll
          for (int d = 0; d < NDT; ++d) o[u][d] *= alpha;
          ol[u] *= alpha;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          u32x4_t pw;
          pw[0] = pack_bf16x2(s[u][2 * st][0], s[u][2 * st][1]);
          pw[1] = pack_bf16x2(s[u][2 * st][2], s[u][2 * st][3]);
          pw[2] = pack_bf16x2(s[u][2 * st + 1][0], s[u][2 * st + 1][1]);
          pw[3] = pack_bf16x2(s[u][2 * st + 1][2], s[u][2 * st + 1][3]);
          pf[u][st] = __builtin_bit_cast(bf16x8_t, pw);
        }
      }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int d = 0; d < NDT; ++d) {
          const u32x2_t a0 = *(const u32x2_t*)(Ks + v_rd[2 * st] + d * 16 * VROW);
          const u32x2_t a1 = *(const u32x2_t*)(Ks + v_rd[2 * st + 1] + d * 16 * VROW);
          const u32x4_t aw = {a0[0], a0[1], a1[0], a1[1]};
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, aw);
          o[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][st], o[0][d], 0, 0, 0);
          o[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][st], o[1][d], 0, 0, 0);
        }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[0][st], ol[0], 0, 0, 0);
        ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[1][st], ol[1], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float l_fin[2] = {rows_sum(ol[0][0]), rows_sum(ol[1][0])};   // the sum sits in the g = 0 lane of the query's column (all lanes take part)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!qvalid[u]) continue;
    const float inv = l_fin[u] > 0.f ? 1.f / l_fin[u] : 0.f;
    bf16_t* op = p.out + ((size_t)b * p.L + (q0 + u * 16 + qi)) * (size_t)(p.nh * HD) + head * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      u32x2_t w;
      w[0] = pack_bf16x2(o[u][d][0] * inv, o[u][d][1] * inv);
      w[1] = pack_bf16x2(o[u][d][2] * inv, o[u][d][3] * inv);
      *(u32x2_t*)(op + 16 * d) = w;
    }
  }
}

// =====================================================================================
// Ping-pong prefill attention (round 3).  Where k_attn_prefill_dma lost its time (in-kernel phase counters):
// per 64-key tile a wave spent 830 cycles on the 24 S^T MFMAs (384 matrix-pipe cycles), 2430 on the softmax (~700 VALU
// cycles) and 1450 on the 28 PV MFMAs (448) -- neither pipe of a SIMD was busier than ~35 %: the SIMD's two waves (one of
// each of the CU's two workgroups) drift in and out of the same phase, share the pipe they both want and leave the other
// idle.  Here the pairing is made deterministic:
//   * ONE workgroup of 8 waves per CU covers 256 queries of a (batch row, head); waves w and w + 4 share a SIMD.  Waves
//     0-3 (group A) and 4-7 (group B) run the same per-tile sequence  [PV(j-1) + S^T(j)] -> [softmax(j)]  half a tile apart:
//     in every step one wave of each SIMD is in its MATRIX phase (52 MFMAs, LDS fragment reads, DMA issue) while its
//     partner is in its VALU phase (softmax of 32 queries x 64 keys); one s_barrier per step keeps them in anti-phase.
//   * VALU work per score cut to the exponential itself: Q arrives PRE-SCALED by scale * log2(e) (the RoPE kernel
//     multiplies before its one rounding to bf16 -- same relative rounding as before), the S^T accumulators START at
//     -m (the running reference of the query's column, lane-uniform over the accumulator rows), so the MFMA result IS
//     the exponent; the reference only moves when a tile's maximum exceeds it by more than 2^8 (wave-uniform slow path
//     that rescales O and l once; P <= 256 in bf16 has the same relative precision) -- and on a query's first visible
//     tile, which sets it.  Fast path per tile and wave: 32 v_exp + 16 v_cvt_pk + ~28 max / select / swap.
//   * K / V^T tiles by LDS-DMA into a ring of four K tiles and five V^T tiles (see issue_batch for the depths), 24 (HD = 96)
//     one-KiB pieces per tile pair = 3 per wave, issued at the top of a VALU phase two to three tiles ahead of their use; counted
//     vmcnt (one batch stays in flight across every barrier), never a full drain inside the loop.  The K tile is staged in a
//     PERMUTED key order that makes every V^T fragment one ds_read_b128 (see load_k), both tiles conflict-free.
//   * fragment reads are taken off the MFMA stream: V^T(j) is read at the top of softmax(j) (lands under the VALU work), K(j+1)
//     at the top of the matrix phase, under the 28 PV MFMAs; the S^T accumulators start from a persistent -reference block.
//   * every tile of K / V^T is fetched ONCE per 256 queries (128 before): half the L2 -> LDS traffic per flop.
// What the compiler must be kept from doing (each cost 1.5-2x, found with SQ_INSTS_VALU and the in-kernel phase timeline,
// tools/attn_pp_timeline.py): conditional updates of the accumulator arrays written as C++ (-> 600 register copies per tile:
// in-place inline asm instead); sinking the exponentials past the asm barrier into the matrix phase (-> results pinned with
// empty volatile asm uses); spilling lane-constant LDS offsets (reload = scratch load = vmcnt(0) = DMA drained: recomputed per
// phase from an opaque lane id).
// Row sums ride on the matrix cores as before (ones-row V^T fragment).
#ifndef P3V_PP_DMA_IN_VALU
#define P3V_PP_DMA_IN_VALU 1
#endif
#ifndef P3V_PP_V_IN_MATRIX
#define P3V_PP_V_IN_MATRIX 1                                    // 1: both halves of the V^T fragments are read in the matrix phase
#endif
#ifdef P3V_PP_DEBUG                                             // tools/attn_pp_timeline.py: per-wave phase timestamps of ONE workgroup
__device__ unsigned long long p3v_ppdbg[8 * 128];
#define PP_T(slot) do { if (blockIdx.x == P3V_PP_DEBUG && (threadIdx.x & 63) == 0 && (slot) < 127) p3v_ppdbg[(threadIdx.x >> 6) * 128 + 1 + (slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int p3v_ppdbg_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_ppdbg), sizeof(unsigned long long) * 8 * 128) == hipSuccess ? 0 : -1;
}
// stamps INSIDE a phase: s_memtime into an SGPR pair (consumed at the end of the step only: no lgkmcnt wait at the stamp)
__device__ unsigned long long p3v_ppdbg2[8 * 64 * 4];
#define PP_S(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(pp_s[k])); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_S_FLUSH(step) do { if (blockIdx.x == P3V_PP_DEBUG && (threadIdx.x & 63) == 0 && (step) < 64) { for (int k_ = 0; k_ < 4; ++k_) p3v_ppdbg2[((threadIdx.x >> 6) * 64 + (step)) * 4 + k_] = pp_s[k_]; } } while (0)
extern "C" int p3v_ppdbg2_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_ppdbg2), sizeof(unsigned long long) * 8 * 64 * 4) == hipSuccess ? 0 : -1;
}
#else
#define PP_T(slot) do { } while (0)
#define PP_S(k) do { } while (0)
#define PP_S_FLUSH(step) do { } while (0)
#endif
template <int HD, bool PRE>
__global__ void __launch_bounds__(512, 1) k_attn_prefill_pp(AttnP p) {
  constexpr int KROW = HD * 2, VROW = 128, NKS = HD / 32, NDT = HD / 16, CPR = HD / 8;
  constexpr bool DV = P3V_PP_DMA_IN_VALU;                      // DMA batches issued from the VALU phase (see issue_batch)
  constexpr int KTILE = 64 * KROW, VTILE = HD * VROW, RING = DV ? 4 : 3, VRING = DV ? 5 : 4;   // ring depths: see issue_batch
  constexpr int NK = KTILE / 1024, NV = VTILE / 1024, NPW = (NK + NV) / 8;   // DMA pieces per tile (K, V^T) and per wave and batch
  constexpr float THR = 8.f;
  static_assert((NK + NV) % 8 == 0, "pieces must divide over 8 waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // K ring [RING][KTILE] | V^T ring [VRING][VTILE]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wave >> 2;
  const int nqb = (p.L + 255) >> 8, per_group = p.head_group * nqb;
  const int hgrp = blockIdx.x / per_group, within = blockIdx.x - hgrp * per_group;
  const int qblk = nqb - 1 - within / p.head_group;             // longest (last) query blocks first
  const int b = blockIdx.z, head = hgrp * p.head_group + within % p.head_group, kvh = head / (p.nh / p.nkv);
  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b / p.pad_div] : 0;
  const int qb0 = qblk * 256, q0 = qb0 + wave * 32;
  const int kv_end = p.causal ? min(total, past + qb0 + 256) : total;
  const int kv_begin = min(pad & ~63, kv_end);
  const int NT = (kv_end - kv_begin + 63) >> 6;
#ifdef P3V_PP_SOLO                                               // timing experiment: group B idles
  const bool active = q0 < p.L && grp == 0;
#else
  const bool active = q0 < p.L;
#endif
  const int wave_last = p.causal ? past + q0 + 31 : total;  // last key position this wave can see
  const float sc2 = p.sc2;                                   // !PRE only
  // interior tiles of this wave: kv0 >= pad, kv0 + 64 <= kv_end, (causal) kv0 + 63 <= past + q0, and no padded query rows
  const int j_int_lo = (past + q0 >= pad) ? ((pad > kv_begin) ? 1 : 0) : (1 << 30);
  const int j_int_hi = min((kv_end - kv_begin) / 64 - 1, p.causal ? (past + q0 - 63 - kv_begin >= 0 ? (past + q0 - 63 - kv_begin) / 64 : -1) : (1 << 30));

  bf16x8_t qf[2][NKS];
  int qpos[2];
  bool qvalid[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qrow = q0 + u * 16 + qi;
    qvalid[u] = qrow < p.L;
    qpos[u] = past + qrow;
    const bf16_t* qp = p.q + (((size_t)b * p.nh + head) * p.L + (qvalid[u] ? qrow : 0)) * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = *(const u32x4_t*)(qp + 32 * ks);
      if (!qvalid[u]) v = (u32x4_t){0, 0, 0, 0};
      qf[u][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const unsigned char* kbase = (const unsigned char*)(p.k_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)p.past_t * HD);
  const unsigned char* vbase = (const unsigned char*)(p.v_past + ((size_t)(b / p.past_div) * p.nkv + kvh) * (size_t)HD * p.past_t);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, 0xffffffff, 0x00020000);

  // ---- this wave's DMA pieces: piece pi = wave + 8 i of the list [K_0 .. K_{NK-1}, V_0 .. V_{NV-1}]
  const unsigned vrow = (unsigned)p.past_t * 2;
  unsigned poff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave + 8 * i;
    if (pi < NK) {
      const int e = pi * 64 + lane, row = e / CPR, pc = e - row * CPR;
      const int sw = HD == 96 ? (-(row >> 2)) & 3 : (row >> 1) & 7;
      const int blk = row >> 4, i_ = row & 15;               // LDS row (block blk, A-operand row i_) holds key pi_key of the tile
      const int key = 32 * (blk >> 1) + 8 * (i_ >> 2) + 4 * (blk & 1) + (i_ & 3);
      poff[i] = key * KROW + ((pc ^ sw) << 4);
    } else {
      const int e = (pi - NK) * 64 + lane, row = e >> 3, pc = e & 7;
      poff[i] = (unsigned)row * vrow + ((pc ^ ((row >> 1) & 7)) << 4);
    }
  }
  // batch jb = K tile jb + 2 and V^T tile jb + 2, issued at the top of the matrix phase M(jb) (group A: step 2 jb, group B:
  // step 2 jb + 1) and waited for by every wave before the barrier that ends step 2 jb + 3.  First readers: K(j) in M(j)
  // (step 2j), V^T(j) in softmax(j) (step 2j + 1) -- both after the end of step 2 (j - 2) + 3.  Last readers of the slot a
  // batch overwrites: K(jb - 1), read in M(jb - 1) (steps 2 jb - 2 / 2 jb - 1) -> three K slots; V^T(jb - 2), read in
  // softmax(jb - 2) (steps 2 jb - 3 / 2 jb - 2) -> FOUR V^T slots (with three, group A's batch jb would land on the tile
  // group B's softmax(jb - 1) is still reading in step 2 jb).  Prologue batches -2, -1 = tiles 0, 1 (clamped into [0, NT)).
  // DV (round 3, late): the matrix phase is the longer one and a DMA issue costs its wave ~100 cycles of MFMA issue, so batch
  // jb >= 1 is issued one step EARLIER, at the top of the wave's VALU phase softmax(jb - 1) (A: step 2 jb - 1, B: 2 jb); the
  // slots it overwrites must then be one tile older: K(jb - 2) (last read in M(jb - 2), steps 2 jb - 4 / 2 jb - 3) -> FOUR
  // K slots, V^T(jb - 3) (second half read in M(jb - 2)) -> FIVE V^T slots (with four, A's batch would land in step 2 jb - 1 on
  // the tile whose second half B's M(jb - 1) reads in that step).  Each wave issues its batches in order (`next_b`); what
  // may stay in flight across the barrier of an odd step is whatever it has issued beyond batch jn - 2 (end_step).
  int next_b = 0;
  auto issue_batch = [&](int jb) {
#ifdef P3V_PP_NODMA                                              // timing experiment: tiles never move
    return;
#endif
    const int tk = min(jb + 2, NT - 1), tv = tk;
    const int ks_ = (jb + 2) % RING, vs_ = (jb + 2) % VRING;
    const int so_k = (kv_begin + 64 * tk) * KROW, so_v = (kv_begin + 64 * tv) * 2;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave + 8 * i;
      const unsigned vo = poff[i];
      if (pi < NK) {
        pf_lptr_t dst = (pf_lptr_t)(smem + ks_ * KTILE + pi * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dst, 16, vo, so_k, 0, 0);
      } else {
        pf_lptr_t dst = (pf_lptr_t)(smem + RING * KTILE + vs_ * VTILE + (pi - NK) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dst, 16, vo, so_v, 0, 0);
      }
    }
  };

  auto issue_at = [&](int ls) {                               // the batch a wave issues at the top of its local step ls, if any
    const int jb = DV ? (ls == 0 ? 0 : (ls & 1) ? (ls + 1) >> 1 : -1) : ((ls & 1) ? -1 : ls >> 1);
    if (jb == next_b && jb <= NT - 3) {
      issue_batch(jb);
      ++next_b;
    }
  };

  // ---- fragment read offsets (as k_attn_prefill_dma), RECOMPUTED at the top of every phase from a lane id the compiler
  // cannot see through: kept live across the tile loop they are spilled (the kernel sits at the 256-register limit of two
  // waves per SIMD), and a spill reload is a scratch load -- its compiler-inserted vmcnt(0) drains the LDS-DMA in flight
  auto lane_now = [&]() {
    unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };

  const u32x4_t ones_w = qi == 0 ? (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : (u32x4_t){0u, 0u, 0u, 0u};
  const bf16x8_t ones_f = __builtin_bit_cast(bf16x8_t, ones_w);
  float m_run[2] = {0.f, 0.f};                               // reference of the exponent (finite always)
  bool unset[2] = {true, true};                              // no visible key seen yet: the next one sets the reference
  f32x4_t ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4_t o[2][NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[u][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t s[2][4];
  bf16x8_t pf[2][2];
#ifdef P3V_PP_DEBUG
  unsigned long long pp_s[4] = {0, 0, 0, 0};
#endif

  // Key order inside a tile.  The S^T accumulator rows a lane holds (rows 4g .. 4g+3 of each 16-row block) become ITS
  // B fragment of the PV product, k index (g, 0..7) = (block 2st', rows 4g..4g+3), (block 2st'+1, rows 4g..4g+3).  The K
  // tile is staged so that A-operand row i of block b is key 32 (b >> 1) + 8 (i >> 2) + 4 (b & 1) + (i & 3) of the tile
  // (the DMA fetches 16-byte chunks from anywhere: a permuted row order is free): a lane's eight k values are then the
  // EIGHT CONSECUTIVE keys 32 st' + 8 g .. + 7, and its V^T fragment is ONE ds_read_b128 (k_attn_prefill_dma: two
  // ds_read_b64 that the compiler fuses into the half-rate ds_read2st64_b64).  Bank conflicts: none on either tile
  // (the 16-lane service groups of ds_read_b128 are {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: the K swizzle for 192-byte
  // rows is chunk ^ (-(row >> 2) & 3) inside aligned groups of four; k_attn_prefill_dma's (row >> 2) & 3 was 2-way
  // conflicted on every read, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50).
  f32x4_t negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // C operand of the S^T products: -reference (PRE) or 0
  bf16x8_t kfr[4][NKS];                                      // K fragments of the tile in work (loaded under the PV MFMAs)
  bf16x8_t vfr[2][NDT];                                      // V^T fragments (loaded under the softmax of the same tile)
  auto load_k = [&](int j) {
    const unsigned char* Ks = smem + (j % RING) * KTILE;
    const unsigned ln = lane_now(), g_ = ln >> 4, qi_ = ln & 15;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const unsigned c = 4 * ks + g_, sw = HD == 96 ? (0u - (qi_ >> 2)) & 3 : (qi_ >> 1) & 7;
      const unsigned off = qi_ * KROW + ((c ^ sw) << 4);
#pragma unroll
      for (int st = 0; st < 4; ++st) kfr[st][ks] = *(const bf16x8_t*)(Ks + off + st * 16 * KROW);
    }
  };
  auto load_v = [&](int j, int st) {                         // half st (keys 32 st .. 32 st + 31) of V^T(j)
    const unsigned char* Vs = smem + RING * KTILE + (j % VRING) * VTILE;
    const unsigned ln = lane_now(), g_ = ln >> 4, qi_ = ln & 15;
    const unsigned off = qi_ * VROW + (((4 * st + g_) ^ ((qi_ >> 1) & 7)) << 4);
#pragma unroll
    for (int d = 0; d < NDT; ++d) vfr[st][d] = *(const bf16x8_t*)(Vs + off + d * 16 * VROW);
  };
  auto qk = [&]() {                                          // S^T(j) = K_j . Q^T (- reference), K_j in kfr
    // the first MFMA of every accumulator takes the persistent block `negm` (-reference in all four rows) as its C
    // operand: no per-tile initialisation of the 32 accumulator registers
    // (k-slice outermost: eight independent accumulators between two MFMAs of one chain -- with the slice innermost a chain's
    //  next MFMA issued two slots after its predecessor, inside its 8-pass latency: SQ_WAIT_INST_ANY was 38 % of the wave cycles)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
#ifdef P3V_PP_NOLDS                                              // timing experiment: no fragment reads
        const bf16x8_t kf = qf[1][ks];
#else
        const bf16x8_t kf = kfr[st][ks];
#endif
        s[0][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], ks == 0 ? negm[0] : s[0][st], 0, 0, 0);
        s[1][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], ks == 0 ? negm[1] : s[1][st], 0, 0, 0);
      }
    }
  };
  auto pv = [&]() {                                          // O^T += V^T_j . P^T(j), l += 1 . P^T(j), V^T_j in vfr
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
#ifdef P3V_PP_NOLDS
        const bf16x8_t vf = qf[0][d % NKS];
#else
        const bf16x8_t vf = vfr[st][d];
#endif
        o[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][st], o[0][d], 0, 0, 0);
        o[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][st], o[1][d], 0, 0, 0);
      }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[0][st], ol[0], 0, 0, 0);
      ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[1][st], ol[1], 0, 0, 0);
    }
  };
  // Every conditional update of the big register arrays (mask, moved reference) is an IN-PLACE inline-asm instruction on
  // the array element ("+v"): written as C++ selects / multiplies the compiler gives the updated array NEW registers and
  // fills the common path with copies -- 600 v_mov per tile and wave in the first version of this kernel (SQ_INSTS_VALU
  // 734 per wave-tile against 130 of real work, profiles/r03_pmc_prefill_attn.txt).
  auto softmax = [&](int j) {
    const int kv0 = kv_begin + 64 * j;
    // every key of the tile visible to every query of the wave -> no per-element mask work (wave-uniform; the tile range
    // [j_int_lo, j_int_hi] is computed once per workgroup: two scalar compares per tile)
    const bool interior = j >= j_int_lo && j <= j_int_hi;
#ifdef P3V_PP_VALUPRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    issue_at(2 * j + 1);                                     // (DV) this wave's pieces of DMA batch j + 1 (later in the phase: no gain / -4 %)
    PP_S(0);
#if !P3V_PP_V_IN_MATRIX
    load_v(j, 0);                                            // first half of V^T(j): lands under the VALU work below (the second
#endif
                                                             // half is read at the top of the matrix phase that consumes it: the
                                                             // VALU phase is the longer of the two)
    float m_t[2];
    if (!PRE) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][st][r] = fmaf(s[u][st][r], sc2, -m_run[u]);
    }
    if (!interior) {
      const float ninf = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = kv0 + 32 * (st >> 1) + 8 * g + 4 * (st & 1) + r;      // key order of the staged tile
            const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos[u]) && qpos[u] >= pad;
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(vis);
            asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(s[u][st][r]) : "v"(ninf), "s"(vm));      // s = vis ? s : -inf
          }
    }
    // (both 16-query halves in ONE basic block: their reduction chains are dependent instruction by instruction, the
    //  scheduler interleaves the two)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifdef P3V_PP_NOMAX
      m_t[u] = s[u][0][0];
      continue;
#endif
      float a = fmaxf(fmaxf(s[u][0][0], s[u][0][1]), s[u][0][2]);      // v_max3 chains
      float c = fmaxf(fmaxf(s[u][0][3], s[u][1][0]), s[u][1][1]);
      a = fmaxf(fmaxf(a, s[u][1][2]), s[u][1][3]);
      c = fmaxf(fmaxf(c, s[u][2][0]), s[u][2][1]);
      a = fmaxf(fmaxf(a, s[u][2][2]), s[u][2][3]);
      c = fmaxf(fmaxf(c, s[u][3][0]), s[u][3][1]);
      a = fmaxf(fmaxf(a, s[u][3][2]), s[u][3][3]);
      m_t[u] = rows_max(fmaxf(a, c));
    }
#ifdef P3V_PP_NOMAX                                              // timing experiment: no reduction, no decision (first tile only)
    const bool slow = unset[0] || unset[1];
#else
    const bool slow = unset[0] || unset[1] || m_t[0] > THR || m_t[1] > THR;
#endif
    PP_S(1);
    if (__builtin_amdgcn_ballot_w64(slow) != 0) {               // wave-uniform: a reference moves (first tile, or a jump > 2^8)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool masked = m_t[u] == -INFINITY;
        const float delta = unset[u] ? (masked ? 0.f : m_t[u]) : fmaxf(m_t[u], 0.f);
        const float alpha = unset[u] ? 1.f : __builtin_amdgcn_exp2f(-delta);
        unset[u] = unset[u] && masked;
        m_run[u] += delta;
        if (PRE) negm[u] = (f32x4_t){-m_run[u], -m_run[u], -m_run[u], -m_run[u]};
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm("v_sub_f32 %0, %0, %1" : "+v"(s[u][st][r]) : "v"(delta));
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %0, %1" : "+v"(o[u][d][r]) : "v"(alpha));
#pragma unroll
        for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, %0, %1" : "+v"(ol[u][r]) : "v"(alpha));
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
#ifdef P3V_PP_NOEXP                                              // timing experiment: no transcendental (single non-packed VALU op instead)
        for (int r = 0; r < 4; ++r) asm("v_mul_f32 %0, 0.5, %0" : "+v"(s[u][st][r]));
#else
        for (int r = 0; r < 4; ++r) s[u][st][r] = __builtin_amdgcn_exp2f(s[u][st][r]);
#endif
      if (u == 1) PP_S(2);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4_t pw;
        pw[0] = pack_bf16x2(s[u][2 * st][0], s[u][2 * st][1]);
        pw[1] = pack_bf16x2(s[u][2 * st][2], s[u][2 * st][3]);
        pw[2] = pack_bf16x2(s[u][2 * st + 1][0], s[u][2 * st + 1][1]);
        pw[3] = pack_bf16x2(s[u][2 * st + 1][2], s[u][2 * st + 1][3]);
        pf[u][st] = __builtin_bit_cast(bf16x8_t, pw);
      }
    }
    // P is USED here as far as the optimiser can tell: without this the exponentials are sunk past the step barrier (an asm
    // barrier orders memory operations only) into the matrix phase that consumes them
    asm volatile("" ::"v"(pf[0][0]), "v"(pf[0][1]), "v"(pf[1][0]), "v"(pf[1][1]));
    PP_S(3);
#ifdef P3V_PP_VALUPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto pin_s = [&]() {                                       // likewise: the S^T MFMAs belong to the phase that issued them
    asm volatile("" ::"v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][3]), "v"(s[1][0]), "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][3]));
  };
  auto pin_o = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("" ::"v"(ol[u]));
#pragma unroll
      for (int d = 0; d < NDT; ++d) asm volatile("" ::"v"(o[u][d]));
    }
  };

  // ---- the step sequence.  Global steps 0 .. 2 NT + 1, one s_barrier after each.  A wave of group g (0 = A, 1 = B) runs
  // its matrix phase M(j) = {issue DMA batch j; PV(j-1); S^T(j)} in step 2j + g and its VALU phase softmax(j) in step
  // 2j + 1 + g.  A wave works on tiles j < NTw (tiles above its diagonal are skipped; monotone in j); the steady-state loop
  // body is STRAIGHT-LINE code over the accumulators (no conditional around pv / qk / softmax: every conditional there costs
  // register copies), the edges are peeled, and a wave that is done keeps issuing its share of the DMA and the barriers.
  int step = 0;
  auto end_step = [&]() {
    // (the scheduler may move anything that is not a memory operation across an asm barrier -- it sank the whole
    //  exponential block of the softmax into the following matrix phase: fence the instruction stream as well)
    __builtin_amdgcn_sched_barrier(0);
    PP_S_FLUSH(step);
    PP_T(2 * step + 1);
    // before the barrier that ends an ODD step every wave has batch jn - 2 (tile jn: read from the next step on) in LDS
    // (group A issued it three steps ago, group B two); the batch issued since (jn - 1, if there was one: the last batch is
    // NT - 3) stays in flight
    if (step & 1) {
      const int jn = (step + 1) >> 1, newer = next_b - (jn - 1);   // batches this wave has issued beyond jn - 2
      if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifndef P3V_PP_NOBAR                                             // (timing experiment: no step barrier)
    asm volatile("s_barrier" ::: "memory");
#endif
    ++step;
    PP_T(2 * step);
    __builtin_amdgcn_sched_barrier(0);
  };
  const int NTw = active ? max(0, min(NT, ((wave_last - kv_begin) >> 6) + 1)) : 0;
  if (NT > 0) {
    issue_batch(-2);
    issue_batch(-1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");          // batch -2 (K tile 0) has landed, batch -1 flies
    asm volatile("s_barrier" ::: "memory");
    if (grp) end_step();                                                // group B runs one step behind group A
    // M(0)
    issue_at(0);
    if (NTw > 0) {
#ifndef P3V_PP_NOPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      load_k(0);
      qk();
      pin_s();
      __builtin_amdgcn_s_setprio(0);
    }
    end_step();
    for (int j = 0; j + 1 < NTw; ++j) {
#ifndef P3V_PP_NOVALU
      softmax(j);
#endif
      end_step();
      issue_at(2 * j + 2);
#ifndef P3V_PP_NOPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      PP_S(0);
#if P3V_PP_V_IN_MATRIX
      load_v(j, 0);
#endif
      load_v(j, 1);                                                     // second half of V^T(j): under the first 12 PV MFMAs
      load_k(j + 1);                                                    // 12 ds_read_b128 in flight under the 28 PV MFMAs (reading them
      __builtin_amdgcn_sched_barrier(0);                                //  behind the first 14 instead, or raising the priority only for
      PP_S(1);                                                          //  the MFMAs: no change / -2 %)
      pv();
      PP_S(2);
      qk();
      pin_o();
      pin_s();
      PP_S(3);
      __builtin_amdgcn_s_setprio(0);
      end_step();
    }
    if (NTw > 0) {                                                      // last tile of this wave: softmax, then PV alone
#ifndef P3V_PP_NOVALU
      softmax(NTw - 1);
#endif
      end_step();
      issue_at(2 * NTw);
#if P3V_PP_V_IN_MATRIX
      load_v(NTw - 1, 0);
#endif
      load_v(NTw - 1, 1);
      pv();
      pin_o();
      end_step();
    }
    while (step <= 2 * NT + 1) {                                        // done (or never had rows): DMA share + barriers only
      const int ls = step - grp;
      if (ls >= 0) issue_at(ls);
      end_step();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float l_fin[2] = {rows_sum(ol[0][0]), rows_sum(ol[1][0])};   // the sum sits in the g = 0 lane of the query's column (all lanes take part)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!qvalid[u]) continue;
    const bool ok = l_fin[u] > 0.f;                                  // a query that is itself padding: 0 (Q7)
    const float inv = ok ? 1.f / l_fin[u] : 0.f;
    bf16_t* op = p.out + ((size_t)b * p.L + (q0 + u * 16 + qi)) * (size_t)(p.nh * HD) + head * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      u32x2_t w;
      w[0] = ok ? pack_bf16x2(o[u][d][0] * inv, o[u][d][1] * inv) : 0u;
      w[1] = ok ? pack_bf16x2(o[u][d][2] * inv, o[u][d][3] * inv) : 0u;
      *(u32x2_t*)(op + 16 * d) = w;
    }
  }
}

// Which prompt-sized kernel (measured, 32 heads x 96, causal, random data; profiles/r03_attn_prefill_kernels.txt; the table the
// launcher's rules come from sits at the rules, in launch_attn_prefill):
//   dma (128 queries, 4 waves, two workgroups per CU): every plain-Q call (the ViT, cached calls) below 3072 tokens, pre-scaled
//       prompts of several rows up to ~3500 tokens -- since it got the deferred-reference softmax: 946 TF/s at 8k (round 2: 606-623);
//   il  (interleaved; 128 or 256 queries): pre-scaled prompts -- ONE row from 768 tokens, several rows from 3584, everything from 6144;
//   pp  (ping-pong, 256 queries): what is left for it -- plain Q or head dim 64 from 3072 tokens.
#include "p3v_attn_il.h"
constexpr int P3V_ATTN_PP_MIN_L = 3072;                          // tools/attn_short_probe.py: dma / pp 65.2 / 67.5 us at 2531, 82.3 / 79.0 at 3072
template <int HD>
static int launch_attn_prefill(const AttnP& p, hipStream_t s) {
  constexpr int LDS = 2 * (64 * (HD * 2 + 16) + HD * (64 * 2 + 16));
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_attn_prefill<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const dim3 grid(p3v_cdiv(p.L, 128), p.nh, p.B);
  if (p.new_is_cache && p.past_t % 64 == 0 && !p3v_tuning().attn_no_dma) {      // every caller in the model
    constexpr int LDS2 = 2 * (64 * HD * 2 + HD * 128);
    static bool attr2_set = false;
    if (!attr2_set) {
      if (hipFuncSetAttribute((const void*)k_attn_prefill_dma<HD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_attn_prefill_dma<HD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess)
        return P3V_ERR_HIP;
      attr2_set = true;
    }
    AttnP q = p;
    const size_t kv_bytes = (size_t)(p.past + p.L) * HD * 4;     // one head's K + V^T
    q.head_group = kv_bytes * p.nh <= (64u << 20) ? p.nh : (p.nh % 8 == 0 && kv_bytes * 8 <= (128u << 20) ? 8 : (p.nh % 4 == 0 ? 4 : p.nh));
    const int pp = p3v_tuning().attn_pp;                         // -1: by shape; 0 / 1: pin (kernel tests run both)
    const int il = p3v_tuning().attn_il;                         // the interleaved kernel (pre-scaled Q only)
    // Which kernel for a pre-scaled prompt (tools/attn_il4_probe.py, 32 heads x 96, causal, us; dma / il 4 waves / il 8 waves):
    //   B = 1:  512: 15.9 / 16.4 / 22.2   768: 21.1 / 20.1 / 26.6   1280: 32.3 / 27.1 / 34.8   2531: 64.6 / 52.6 / 58.8
    //           3072: 81.4 / 74.6 / 69.5   4096: 119 / 114 / 108    8192: 436 / 403 / 377
    //   B = 2:  1280: 40.0 / 36.5 / 40.3   2531: 91.8 / 99.8 / 106   4096: 221 / 219 / 228   5120: 343 / 320 / 328
    //   B = 4:  2531: 168 / 183 / 190      3072: 252 / 254 / 271   4096: 426 / 410 / 418   6144: 939 / 886 / 893   8192: 1614 / 1504 / 1477
    //   B = 8:  1280: 112 / 127 / 142      2531: 385 / 396 / 414
    // -> long prompts: 256-query workgroups (half the L2 -> LDS bytes per flop); ONE row of a few thousand tokens: 128-query
    //    workgroups of the interleaved kernel (best balance, two workgroups per CU); several rows: the 128-query kernel up to ~3500 tokens.
    const bool il_auto = p.L >= 6144 || (p.B == 1 ? p.L >= 768 : p.L >= 3584);
    if (p.q_prescaled && (il > 0 || (il < 0 && pp != 0 && il_auto))) {   // (pp = 0 pins the dma kernel)
      constexpr int LDS4 = 3 * 64 * HD * 2 + 3 * HD * 128;
      static bool attr4_set = false;
      if (!attr4_set) {
        if (hipFuncSetAttribute((const void*)k_attn_prefill_il<HD, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_attn_prefill_il<HD, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4) != hipSuccess)
          return P3V_ERR_HIP;
        attr4_set = true;
      }
      // 128-query workgroups (two per CU) while 256-query ones would leave the launch bounded by its longest block
      const int nw = p3v_tuning().attn_il_waves;
      const bool small = nw == 4 || (nw < 0 && p.L < (p.B == 1 ? 2816 : 6144));
      if (small) hipLaunchKernelGGL((k_attn_prefill_il<HD, 4>), dim3(p.nh * p3v_cdiv(p.L, 128), 1, p.B), dim3(256), LDS4, s, q);
      else hipLaunchKernelGGL((k_attn_prefill_il<HD, 8>), dim3(p.nh * p3v_cdiv(p.L, 256), 1, p.B), dim3(512), LDS4, s, q);
      P3V_CHECK_LAUNCH();
      return P3V_OK;
    }
    if (pp > 0 || (pp < 0 && p.L >= P3V_ATTN_PP_MIN_L && (p.q_prescaled || HD == 64))) {   // (<96, plain>: 5 spilled registers, tests only)
      constexpr int LDS3 = (P3V_PP_DMA_IN_VALU ? 4 : 3) * 64 * HD * 2 + (P3V_PP_DMA_IN_VALU ? 5 : 4) * HD * 128;   // K ring + V^T ring (issue_batch)
      static bool attr3_set = false;
      if (!attr3_set) {
        if (hipFuncSetAttribute((const void*)k_attn_prefill_pp<HD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_attn_prefill_pp<HD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3) != hipSuccess)
          return P3V_ERR_HIP;
        attr3_set = true;
      }
      const dim3 grid3(p.nh * p3v_cdiv(p.L, 256), 1, p.B);
      if (p.q_prescaled) hipLaunchKernelGGL((k_attn_prefill_pp<HD, true>), grid3, dim3(512), LDS3, s, q);
      else hipLaunchKernelGGL((k_attn_prefill_pp<HD, false>), grid3, dim3(512), LDS3, s, q);
      P3V_CHECK_LAUNCH();
      return P3V_OK;
    }
    if (p.q_prescaled) hipLaunchKernelGGL((k_attn_prefill_dma<HD, true>), dim3(p.nh * p3v_cdiv(p.L, 128), 1, p.B), dim3(256), LDS2, s, q);
    else hipLaunchKernelGGL((k_attn_prefill_dma<HD, false>), dim3(p.nh * p3v_cdiv(p.L, 128), 1, p.B), dim3(256), LDS2, s, q);
  } else {
    hipLaunchKernelGGL(k_attn_prefill<HD>, grid, dim3(256), LDS, s, p);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}


extern "C" int64_t p3v_attention_ws_bytes(int B, int L, int n_heads, int hd, int n_split) {
  if (L > P3V_DECODE_MAX_L || n_split < 1) return 0;
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
  p.q_prescaled = a->q_prescaled != 0;
  p.sc2 = p.q_prescaled ? 1.f : a->scale * 1.4426950408889634f;
  p.n_split = a->n_split;
  p.split_mode = (a->L <= P3V_DECODE_MAX_L && a->n_split > 1) ? 1 : 0;
  if (p.split_mode && !a->ws) return P3V_ERR_ARG;
  if (!p.split_mode) p.n_split = 1;
  hipStream_t s = (hipStream_t)stream;
  if (!p.split_mode && a->L > P3V_DECODE_MAX_L && !p3v_tuning().attn_old)
    return a->hd == 96 ? launch_attn_prefill<96>(p, s) : launch_attn_prefill<64>(p, s);
  dim3 grid(p.split_mode ? p.n_split : p3v_cdiv(a->L, 64), a->n_heads, a->B);
  if (a->hd == 96) hipLaunchKernelGGL(k_attn<96>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_attn<64>, grid, dim3(256), 0, s, p);
  P3V_CHECK_LAUNCH();
  if (p.split_mode) {
    hipLaunchKernelGGL(k_attn_combine2, dim3(a->B * a->n_heads * a->L), dim3(combine_threads(a->n_split)), 0, s, a->ws, a->out, a->L, a->n_heads,
                       a->hd, p.n_split);
    P3V_CHECK_LAUNCH();
  }
  return P3V_OK;
}

// =====================================================================================
// Fused decode step attention (L <= 16 new tokens, n_beam == 1):
//   split(qkv) + _rotate_half + KV append + split-KV attention   (phi.py:443-457, 542-548)
// in ONE launch, one WAVE per workgroup (no block barriers), grid (n_split, heads, B).
//   * Q is rotated in registers straight into MFMA B fragments;
//   * keys/values of positions < past stream from the cache (bf16, 16-byte loads,
//     24 in flight per lane per 64-key tile); the L new positions are rotated from
//     the qkv row on the fly by the one workgroup whose key range covers them, which
//     also appends them to the cache (so no other workgroup ever reads rows that are
//     written in this launch);
//   * partial (m, l, O) per split go to `ws`; `k_attn_combine2` merges them.
// HBM-bound: algorithmic bytes = 2 * past * n_kv * hd * 2 per (batch row, layer).
struct AttnDecP {
  const bf16_t* qkv; const float* cos_t; const float* sin_t; bf16_t* k_cache; bf16_t* v_cache;
  const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, nh, nkv, past, cache_t, rope_bstride, n_split;   // cos/sin row of (b, new position r) = b*rope_bstride + r
  float scale;
  int chunk, grp, grp_magic;   // host-side: keys per split (multiple of 64), heads per kv head and ceil(2^16 / grp)
  int merge;                   // nonzero: the last split of a (b, head) merges the partials itself (split_merge)
  bf16_t* out;                 // [B, L, nh * 96] (fused merge only)
  // fused o_proj + residual (k_attn_decode128_o only; B = L = 1, nh * 96 = 3072): see attn_decode_body128<true>
  const void* o_w;             // o_proj weight [o_n, nh * 96]: bf16, or 4-bit group-64 (device layout, with o_sb) in k_attn_decode128_o4
  const void* o_sb;            //   its scale | bias words [o_n, nh * 96 / 64] (4-bit only)
  bf16_t* o_x;                 // residual stream row [o_n]: x += bf16(W_o . attention output), in place
  bf16_t* o_rearm;             // the OTHER attention-output buffer: set to the all-ones sentinel by this launch
  int o_n;
  int fo_remap;                // fused launch: 1-D grid with the roles placed by virtual CU (fo_map) instead of the (split, head) grid
};

// ---- fused split-KV merge ("last workgroup merges") without flags.  `ws` holds the SENTINEL bit pattern (all ones: a
// NaN no arithmetic produces) in every word between launches.  A split stores its partial with write-through stores and is
// done -- no store wait, no flag.  The workgroup of the highest split of a (b, head) polls the partial WORDS themselves
// (4-byte stores are atomic, so a word is either the sentinel or final; 24 loads in flight per lane), merges when none is
// the sentinel, and puts the sentinel back.  Its own partial never leaves LDS (`own`).  Compared with data + ready flags
// this takes the store acknowledgement, the flag store and a flag poll off the critical path of every launch (tail of the
// launch at 21 splits: 3.3 -> 2.4 us).  No cache write-back / invalidate fences anywhere (an agent-scope release costs
// ~20 us on this chip, tools/scratch/persist_gemv.hip): partials are stored and loaded with agent-scope relaxed atomics
// (write-through / cache-bypassing accesses).  Workgroups are dispatched in linear order, so by the time the merging
// workgroup runs every other split of its (b, head) is resident or finished and the wait cannot deadlock; it is bounded
// anyway, and when the bound runs out the output is NaN so that the failure is loud (api._rows raises) instead of a
// silently wrong token.
#define WS_SENTINEL 0xffffffffu
#ifdef P3V_ATTN_TIMING                                         // tools/attn_timeline.py: 100 MHz timestamps per workgroup
__device__ long long p3v_tbuf[8192 * 16];
#define TMARK(k) do { if (threadIdx.x == 0) p3v_tbuf[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = wall_clock64(); } while (0)
#define TVAL(k, v) do { if (threadIdx.x == 0) p3v_tbuf[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = (v); } while (0)
extern "C" int p3v_timing_read(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_tbuf), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#else
#define TMARK(k)
#define TVAL(k, v)
#endif
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool is_sentinel(float v) { return __builtin_bit_cast(uint32_t, v) == WS_SENTINEL; }
// Round 6: 16- and 8-byte write-through stores for the sentinels a merge puts back (24 x 16 bytes + 8 per record instead of 98 dword
// stores: a write-through `sc1` store leaves the CU as one fabric write per lane whatever its width, MI355X_MICROARCH.md).  The same
// widening of the partial records and of the merged row was built and measured SLOWER (+25 us per decode step: the wide stores of a
// record become visible to the polling merger ~0.5 us later; profiles/r06_attn_oproj_placement.txt), so those stay dword stores.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_wt4(void* p, f32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_wt2(void* p, f32x2_t v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// merge of the n_split partials of one (b, head) by the calling workgroup (NT threads = G groups of 64 lanes).
// own: the calling workgroup's partial (split n_split - 1) in LDS, [16][HD + 2] like a `ws` record, or nullptr when it
// went to `ws` like the others.
#define SPLIT_MERGE_SCRATCH(NT) ((NT) / 64 * 130 + 2)          // floats: pm[G], pl[G], part[G][128], timeout flag
template <int NT>
__device__ __forceinline__ void split_merge(float* base0, bf16_t* out0, size_t out_qstride, int L, int n_split,
                                            const float* own, float* scratch, bool wt_out = false) {
  constexpr int HD = 96, G = NT / 64;
  float* pm = scratch;
  float* pl = scratch + G;
  float (*part)[128] = (float (*)[128])(scratch + 2 * G);
  int* s_timeout = (int*)(scratch + 2 * G + G * 128);
  const int t = threadIdx.x, grp = t >> 6, d0 = t & 63;
  const size_t sstr = (size_t)16 * (HD + 2);
  const bool two = d0 + 64 < HD;
  const float sentinel = __builtin_bit_cast(float, WS_SENTINEL);
  if (t == 0) *s_timeout = 0;
  __syncthreads();                                             // also: `own` is complete
  for (int q = 0; q < L; ++q) {
    float* base = base0 + (size_t)q * (HD + 2);
    const float* own_q = own ? own + q * (HD + 2) : nullptr;
    const int n_glob = own_q ? n_split - 1 : n_split;          // partials that come through `ws`
    float m, l, a0, a1;
    auto fold = [&](float ms, float ls, float o0, float o1) {
      const float mn = fmaxf(m, ms);
      const float mu = mn == -INFINITY ? 0.f : mn;
      const float ca = __builtin_amdgcn_exp2f(m - mu), cb = __builtin_amdgcn_exp2f(ms - mu);
      // one rounded product + one fma per accumulator, written out: this function is inlined into several kernels (merge only,
      // merge + o_proj, int8 KV) whose results are compared bit for bit, and hipcc contracts `x * a + y * b` differently from one
      // instantiation to the next (tools/stress_fused_oproj.py: one bf16 ulp in one word of ~4 % of random inputs)
      float lb = ls * cb, b0 = o0 * cb, b1 = o1 * cb;
      asm volatile("" : "+v"(lb), "+v"(b0), "+v"(b1));
      l = __builtin_fmaf(l, ca, lb);
      a0 = __builtin_fmaf(a0, ca, b0);
      a1 = __builtin_fmaf(a1, ca, b1);
      m = mn;
    };
    constexpr int UB = 6;                                      // partials per load round: 4 * UB loads in flight per lane
    for (unsigned tries = 0;; ++tries) {
      m = -INFINITY, l = 0.f, a0 = 0.f, a1 = 0.f;
      bool bad = false;
      for (int s = grp; s < n_glob; s += UB * G) {
        float ms[UB], ls[UB], o0[UB], o1[UB];
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          const float* r = base + (size_t)min(s + j * G, n_glob - 1) * sstr;
          ms[j] = ld_wt(r + HD); ls[j] = ld_wt(r + HD + 1);
          o0[j] = ld_wt(r + d0); o1[j] = two ? ld_wt(r + d0 + 64) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          if (s + j * G < n_glob) {
            bad |= is_sentinel(ms[j]) | is_sentinel(ls[j]) | is_sentinel(o0[j]) | is_sentinel(o1[j]);
            fold(ms[j], ls[j], o0[j], o1[j]);
          }
        }
      }
      if (!__any(bad)) { TVAL(6, (long long)tries); break; }
      if (tries >= (1u << 18)) { *s_timeout = 1; break; }
    }
    TMARK(4);
    if (own_q && grp == (n_split - 1) % G)                     // split n_split - 1 is the last one of its group: same order as through `ws`
      fold(own_q[HD], own_q[HD + 1], own_q[d0], two ? own_q[d0 + 64] : 0.f);
    if (G > 1) {
      __syncthreads();
      if (d0 == 0) { pm[grp] = m; pl[grp] = l; }
      part[grp][d0] = a0;
      if (two) part[grp][d0 + 64] = a1;
      __syncthreads();
      if (wt_out) {
        // consumers in the SAME launch poll these words (fused o_proj): same arithmetic per element, two elements per thread,
        // one write-through 4-byte store.  (Writing four copies of the row for the consumers to spread over -- they all re-read
        // it past the caches -- changed nothing measurable: tools/attn_o_timeline.py.)
        if (t < HD / 2) {
          const bool poison = *s_timeout != 0;
          float M = pm[0];
#pragma unroll
          for (int k = 1; k < G; ++k) M = fmaxf(M, pm[k]);
          const float Mu = M == -INFINITY ? 0.f : M;
          float acc0 = 0.f, acc1 = 0.f, lsum = 0.f;
#pragma unroll
          for (int k = 0; k < G; ++k) {
            const float c = __builtin_amdgcn_exp2f(pm[k] - Mu);
            acc0 = __builtin_fmaf(c, part[k][2 * t], acc0);
            acc1 = __builtin_fmaf(c, part[k][2 * t + 1], acc1);
            lsum = __builtin_fmaf(c, pl[k], lsum);
          }
          const uint32_t wv = poison ? 0x7fc07fc0u : pack_bf16x2(lsum > 0.f ? acc0 / lsum : 0.f, lsum > 0.f ? acc1 / lsum : 0.f);
          __hip_atomic_store((uint32_t*)(out0 + (size_t)q * out_qstride) + t, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (t < HD) {
        const bool poison = *s_timeout != 0;
        float M = pm[0];
#pragma unroll
        for (int k = 1; k < G; ++k) M = fmaxf(M, pm[k]);
        const float Mu = M == -INFINITY ? 0.f : M;
        float acc = 0.f, lsum = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) {
          const float c = __builtin_amdgcn_exp2f(pm[k] - Mu);
          acc = __builtin_fmaf(c, part[k][t], acc);
          lsum = __builtin_fmaf(c, pl[k], lsum);
        }
        out0[(size_t)q * out_qstride + t] = poison ? (bf16_t)0x7fc0 : f32_to_bf16(lsum > 0.f ? acc / lsum : 0.f);
      }
    } else {
      const float inv = *s_timeout ? __builtin_nanf("") : (l > 0.f ? 1.f / l : 0.f);
      out0[(size_t)q * out_qstride + d0] = f32_to_bf16(a0 * inv);
      if (two) out0[(size_t)q * out_qstride + d0 + 64] = f32_to_bf16(a1 * inv);
    }
    TMARK(5);
    for (int s = grp; s < n_glob; s += G) {                    // sentinel back (after the output: off its path): ready for the next launch
      if ((q & 1) == 0) {                                      // (even rows of a record are 16-byte aligned: 24 x 16 bytes + 8)
        if (d0 < 24) st_wt4(base + s * sstr + 4 * d0, (f32x4_t){sentinel, sentinel, sentinel, sentinel});
        else if (d0 == 24) st_wt2(base + s * sstr + HD, (f32x2_t){sentinel, sentinel});
      } else {
        st_wt(base + s * sstr + d0, sentinel);
        if (two) st_wt(base + s * sstr + d0 + 64, sentinel);
        if (d0 == 0) { st_wt(base + s * sstr + HD, sentinel); st_wt(base + s * sstr + HD + 1, sentinel); }
      }
    }
  }
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  const f16x2_t r = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, r);
}

// operands of one rotated 8-wide chunk, split into a load half and a math half so that the loads can be issued
// ahead of other memory traffic and consumed later
struct RopeRaw { u32x4_t x0, x1; float4 c0, c1, s0, s1; };
__device__ __forceinline__ RopeRaw rope_fetch(const bf16_t* head_row, int c, const float* ct, const float* st) {
  constexpr int HALF = 48;
  const int d0 = c * 8, lo = d0 < HALF, tb = lo ? d0 : d0 - HALF;
  RopeRaw r;
  r.x0 = *(const u32x4_t*)(head_row + d0);
  r.x1 = *(const u32x4_t*)(head_row + (lo ? d0 + HALF : d0 - HALF));
  r.c0 = *(const float4*)(ct + tb); r.c1 = *(const float4*)(ct + tb + 4);
  r.s0 = *(const float4*)(st + tb); r.s1 = *(const float4*)(st + tb + 4);
  return r;
}
template <bool F16 = false>
__device__ __forceinline__ u32x4_t rope_apply(const RopeRaw& r, int c, unsigned sh16 = 16, unsigned himask = 0xffff0000u) {
  auto bf16lo = [&](unsigned x) { return __builtin_bit_cast(float, x << sh16); };
  auto bf16hi = [&](unsigned x) { return __builtin_bit_cast(float, x & himask); };
  const float cs[8] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w, r.c1.x, r.c1.y, r.c1.z, r.c1.w};
  const float sn[8] = {r.s0.x, r.s0.y, r.s0.z, r.s0.w, r.s1.x, r.s1.y, r.s1.z, r.s1.w};
  const float sg = c < 6 ? -1.f : 1.f;
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v0 = bf16lo(r.x0[j]) * cs[2 * j] + sg * bf16lo(r.x1[j]) * sn[2 * j];
    const float v1 = bf16hi(r.x0[j]) * cs[2 * j + 1] + sg * bf16hi(r.x1[j]) * sn[2 * j + 1];
    o[j] = F16 ? pack_f16x2(v0, v1) : pack_bf16x2(v0, v1);
  }
  return o;
}

template <bool F16 = false>
__device__ __forceinline__ u32x4_t rope_chunk(const bf16_t* head_row, int c, const float* ct, const float* st) {
  // rotated 8-wide chunk c (0..11) of a 96-wide head: low half pairs with +48, high half with -48
  constexpr int HALF = 48;
  const int d0 = c * 8, lo = d0 < HALF;
  const u32x4_t x0 = *(const u32x4_t*)(head_row + d0);
  const u32x4_t x1 = *(const u32x4_t*)(head_row + (lo ? d0 + HALF : d0 - HALF));
  const int tb = lo ? d0 : d0 - HALF;
  const float4 c0 = *(const float4*)(ct + tb), c1 = *(const float4*)(ct + tb + 4);
  const float4 s0 = *(const float4*)(st + tb), s1 = *(const float4*)(st + tb + 4);
  const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float sg = lo ? -1.f : 1.f;
  u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v0 = bf16lo(x0[j]) * cs[2 * j] + sg * bf16lo(x1[j]) * sn[2 * j];
    const float v1 = bf16hi(x0[j]) * cs[2 * j + 1] + sg * bf16hi(x1[j]) * sn[2 * j + 1];
    o[j] = F16 ? pack_f16x2(v0, v1) : pack_bf16x2(v0, v1);
  }
  return o;
}

// cache policy of the decode kernels' K / V^T tile DMA: 2 = nt.  A decode step reads every cache tile exactly once, so the
// lines need not displace the weights' in L2 / MALL: measured on the bench step 1.856 -> 1.815 ms (-1.3 us per layer);
// (the prefill GEMMs keep the default policy: every CU re-reads the same weight slices there)
#ifndef P3V_ATTN_AUX
#define P3V_ATTN_AUX 2
#endif
typedef const __attribute__((address_space(1))) void* dec_gptr_t;
typedef __attribute__((address_space(3))) void* dec_lptr_t;

// Workgroup = 4 waves x one 64-key tile; wave w owns keys [16w, 16w+16) of the tile end to end: it DMAs its own K
// slice (16 rows x 192 B, contiguous in the cache) and a quarter of the V^T tile BY ROWS (24 rows x 128 B, whole cache
// lines; the tile is shared after the workgroup barrier) HBM -> LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write pass), runs S^T = K.Q^T (3 MFMA 16x16x32), a
// 4-value-per-lane softmax and O^T += V^T.P^T (6 MFMA 16x16x16, P straight from the S^T accumulator layout), and
// the four wave partials are merged through LDS before one (m, l, O) partial per workgroup goes to `ws`.
// Why 4 short waves instead of 1 long one: a lone wave issues one dependent instruction per ~8 cycles, so the
// duration of a decode-attention launch is the instruction count of its longest wave, not its bytes.
// The DMA destination is lane-linear, so the bank-conflict swizzle is applied to the SOURCE chunk index and again
// on the ds_read side:
//   K  : 16-B chunk c (0..11) of row r lives at chunk c ^ ((r >> 2) & 3)      (192-B rows: rows r, r+4 share banks)
//   V^T: 16-B chunk c (0..1)  of row d lives at chunk c ^ ((d >> 3) & 1)      (32-B rows: rows d, d+8 share banks)
//   Q   (rotated by the first 12 L threads, [16 queries][96]): same swizzle as K.
// The launcher guarantees one tile per workgroup (n_split * 64 >= cache_t); longer splits take k_attn_decode_stream.
#ifdef P3V_ATTN_DEBUG
__device__ unsigned long long p3v_dbg[16];
#define DBG_T(i) do { if (blockIdx.x == 7 && blockIdx.y == 3 && threadIdx.x == 0) p3v_dbg[i] = __builtin_readcyclecounter(); } while (0)
// launch-level timeline on the 100 MHz wall clock: first workgroup's entry, and entry / partials stored / merge done of the
// LAST workgroup of the launch (highest split of the last head = the last merger)
#define DBG_W(i, first) do { if (threadIdx.x == 0 && ((first) ? (blockIdx.x == 0 && blockIdx.y == 0) : (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))) p3v_dbg[i] = wall_clock64(); } while (0)
#else
#define DBG_T(i)
#define DBG_W(i, first)
#endif
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void attn_decode_body(const AttnDecP& p, const int bx, const int by, const int bz, unsigned char* KV) {
  DBG_T(0);
  DBG_W(10, true);
  DBG_W(12, false);
  constexpr int TK = 64, WK = 16, HD = 96, KROW = HD * 2, VROW = WK * 2, NKS = 3, NDT = 6, CPR = 12;
  constexpr int KS_BYTES = WK * KROW, VS_BYTES = HD * VROW, WREG = KS_BYTES + VS_BYTES;
  static_assert(WREG >= 16 * HD * 4, "a wave's O partial reuses its tile region");
  __shared__ __attribute__((aligned(16))) unsigned char Qs[16 * KROW];
  __shared__ float Ml[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = bz, head = by, kvh = (head * p.grp_magic) >> 16;      // head / grp without a divide
  const bool kv_writer = head == kvh * p.grp;
  unsigned char* wreg = KV + wave * WREG;

  // ---- the cache length and this row's left padding: scalar loads issued first, consumed after everything
  //      that does not depend on them has been put in flight
  int past = p.past, pad = 0;
  if (p.d_past) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(past) : "s"(p.d_past) : "memory");
  if (p.pad_len) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(pad) : "s"(p.pad_len + b) : "memory");

  // ---- tile DMA.  The key range of a split is STATIC (a function of the cache capacity, not of the current
  //      length), so its first tile is requested before the cache length has arrived.  K: LDS slot i = j*64 + lane
  //      holds (row i/12, physical chunk i%12).  V^T: slot i -> (row i/2, physical chunk i%2).  Rows beyond the
  //      live length are fetched too (allocated, finite) and masked.
  const unsigned char* kc = (const unsigned char*)(p.k_cache + ((size_t)b * p.nkv + kvh) * (size_t)p.cache_t * HD);
  bf16_t* vc = p.v_cache + ((size_t)b * p.nkv + kvh) * (size_t)HD * p.cache_t;          // V^T: [hd][cache_t]
  const int kv_lo = bx * TK, kv_hi = min(p.cache_t, kv_lo + TK);
  unsigned koff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = j * 64 + lane, r0 = i / CPR, pc = i - r0 * CPR;
    koff[j] = r0 * KROW + ((pc ^ ((r0 >> 2) & 3)) << 4);
  }
  const size_t vrow = (size_t)p.cache_t * 2;                   // bytes per V^T row
  // LDS image: [K slice of wave 0..3 (16 keys x 192 B each)] [V^T tile: 96 rows x 128 B].  K is sliced by KEYS (wave w
  // fetches and consumes keys 16w..16w+15: 3 KiB contiguous in the cache).  V^T is fetched by ROWS -- wave w brings rows
  // 24w..24w+23 across all 64 keys, eight whole 128-byte lines per DMA instruction -- and consumed by keys after the
  // workgroup barrier that follows the DMA anyway: a per-wave 16-key slice of V^T would be 96 rows x 32 bytes, quarter lines
  // per request (half / quarter-line streams run at 4.0-4.3 TB/s on this chip, whole lines at 5.5:
  // tools/scratch/frag_stream.hip).  16-byte chunk c of row d sits in slot c ^ ((d >> 1) & 7): the 16 rows x 2 key groups of
  // a ds_read_b64 lane group then hit 32 different banks.
  unsigned char* kslice = KV + wave * KS_BYTES;
  unsigned char* vtile = KV + 4 * KS_BYTES;
  auto load_tile = [&](int kv0) {
    const unsigned char* ksrc = kc + (size_t)(kv0 + WK * wave) * KROW;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      __builtin_amdgcn_global_load_lds((dec_gptr_t)(ksrc + koff[j]), (dec_lptr_t)(kslice + j * 1024), 16, 0, P3V_ATTN_AUX);
    const unsigned char* vs = (const unsigned char*)vc + (size_t)kv0 * 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int d = 24 * wave + 8 * j + (lane >> 3);
      const unsigned voff = (unsigned)d * (unsigned)vrow + ((((unsigned)lane & 7) ^ (((unsigned)d >> 1) & 7)) << 4);
      __builtin_amdgcn_global_load_lds((dec_gptr_t)(vs + voff), (dec_lptr_t)(vtile + wave * VS_BYTES + j * 1024), 16, 0, P3V_ATTN_AUX);
    }
  };
  load_tile(min(kv_lo, p.cache_t - TK));                    // unconditional (an empty split fetches a tile it never uses)
  DBG_T(9);

  // ---- the L new rows: thread (r, c) = tid / 12, tid % 12 < L rotates chunk c of row r, Q for this head and K
  //      (kept in registers until the tile that holds position past + r has landed); thread tid < L * 96 also
  //      fetches one element of the new V rows.  All of it is requested now, under the DMA.
  const int row_w = (p.nh + 2 * p.nkv) * HD;                   // qkv row width
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);
  const int tr = tid / CPR, tc = tid - tr * CPR;
  const bool rtask = tr < p.L;
  RopeRaw qraw, kraw;
  if (rtask) {
    const bf16_t* row = p.qkv + ((size_t)b * p.L + tr) * row_w;
    qraw = rope_fetch(row + head * HD, tc, cos_b + tr * (HD / 2), sin_b + tr * (HD / 2));
    const bf16_t* krow = row + (p.nh + kvh) * HD;
    kraw = qraw;
    kraw.x0 = *(const u32x4_t*)(krow + tc * 8);
    kraw.x1 = *(const u32x4_t*)(krow + (tc < 6 ? tc * 8 + 48 : tc * 8 - 48));
  }
  const bf16_t* vnew = p.qkv + (size_t)b * p.L * row_w + (p.nh + p.nkv + kvh) * HD;   // + r * row_w + d
  const int n_vnew = p.L * HD;
  bf16_t v_early = 0;
  if (tid < n_vnew) {
    const int r = tid / HD;
    v_early = vnew[(size_t)r * row_w + (tid - r * HD)];
  }
  if (tid < 16 * CPR) {                                        // rotated Q (zero rows for q >= L) -> LDS
    u32x4_t v = {0, 0, 0, 0};
    if (rtask) v = rope_apply(qraw, tc);
    *(u32x4_t*)(Qs + tr * KROW + ((tc ^ ((tr >> 2) & 3)) << 4)) = v;
  }
  DBG_T(2);

  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(past), "+s"(pad)::"memory");
  DBG_T(1);
  const int total = past + p.L;
  const int kv_end = min(total, kv_hi);
  const int qpos = past + qi;
  const bool qvalid = qi < p.L;
  const float sc2 = p.scale * 1.4426950408889634f;            // scale * log2(e): softmax runs on exp2

  // ---- fragment read offsets (swizzled as above)
  const unsigned k_rd = qi * KROW + ((g ^ ((qi >> 2) & 3)) << 4);                         // + ks*64
  const unsigned v_rd = qi * 128 + (((2 * wave + (g >> 1)) ^ ((qi >> 1) & 7)) << 4) + (g & 1) * 8;   // + dt*16*128 (row 16*dt + qi)

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t qf[NKS];

  const int kv0 = kv_lo;
  if (kv0 < kv_end) {
    const unsigned char* Wb = kslice;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the DMA of this wave's slices has landed
    __syncthreads();                                           // Qs is complete, every wave's slices are in LDS
    DBG_T(3);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8_t*)(Qs + k_rd + ks * 64);
    if (kv0 + TK > past) {                                     // workgroup-uniform: the tile holds new positions
      // new positions in this tile: K rotated from the qkv row, V copied, into the LDS slices of the waves that
      // own them, and appended to the cache by the writer workgroup (no other workgroup ever reads cache rows
      // that are written in this launch)
      if (rtask) {
        const int t = past + tr, rr = t - kv0;
        if (rr >= 0 && rr < TK && t < kv_end) {
          const u32x4_t kn = rope_apply(kraw, tc);
          const int row = rr & 15;
          *(u32x4_t*)(KV + (rr >> 4) * KS_BYTES + row * KROW + ((tc ^ ((row >> 2) & 3)) << 4)) = kn;
          if (kv_writer) *(u32x4_t*)(p.k_cache + (((size_t)b * p.nkv + kvh) * p.cache_t + t) * HD + tc * 8) = kn;   // phi.py:545
        }
      }
#pragma unroll 1
      for (int idx = tid; idx < n_vnew; idx += 256) {
        const int r = idx / HD, d = idx - r * HD, t = past + r, rr = t - kv0;
        if (rr >= 0 && rr < TK && t < kv_end) {
          const bf16_t val = idx == tid ? v_early : vnew[(size_t)r * row_w + d];
          *(bf16_t*)(vtile + d * 128 + (((rr >> 3) ^ ((d >> 1) & 7)) << 4) + (rr & 7) * 2) = val;
          if (kv_writer) vc[(size_t)d * p.cache_t + t] = val;                                               // phi.py:546
        }
      }
      __syncthreads();
    }

    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8_t kf = *(const bf16x8_t*)(Wb + k_rd + ks * 64);
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);
    }
    DBG_T(4);
    float m_t = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = kv0 + WK * wave + 4 * g + r;
      const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
      s[r] = vis ? s[r] * sc2 : -INFINITY;                     // log2 domain: scale*log2(e) folded in
      m_t = fmaxf(m_t, s[r]);
    }
    m_t = rows_max(m_t);
    const float m_new = fmaxf(m_run, m_t);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    float l_t = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
      l_t += s[r];
    }
    l_run = rows_sum(l_t);
    m_run = m_new;
    DBG_T(5);
    const u32x2_t pw = {pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
    const s16x4_t pf = __builtin_bit_cast(s16x4_t, pw);
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      const s16x4_t vf = *(const s16x4_t*)(vtile + v_rd + d * 16 * 128);
      o[d] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pf, o[d], 0, 0, 0);
    }
    DBG_T(6);
  }

  // ---- merge the four wave partials: each wave parks (O, m, l) of its valid queries in its own (now dead) tile
  //      region, then thread idx < L*96 folds element (q, d) over the waves and writes the workgroup partial
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // no DMA may still target this region / LDS at exit
  __syncthreads();                                            // the V^T tile is shared: every wave is done with it before it is reused
  if (qvalid) {
    float* Ow = (float*)wreg + qi * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d) *(f32x4_t*)(Ow + 16 * d + 4 * g) = o[d];
    if (g == 0) { Ml[wave][qi][0] = m_run; Ml[wave][qi][1] = l_run; }
  }
  __syncthreads();
#pragma unroll 1
  for (int idx = tid; idx < n_vnew; idx += 256) {
    const int q = idx / HD, d = idx - q * HD;
    float mk[4], M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mk[k] = Ml[k][q][0]; M = fmaxf(M, mk[k]); }
    const float Mu = M == -INFINITY ? 0.f : M;
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = __builtin_amdgcn_exp2f(mk[k] - Mu);
      acc += c * ((const float*)(KV + k * WREG))[q * HD + d];
      lsum += c * Ml[k][q][1];
    }
    float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + bx) * 16 + q) * (HD + 2);
    st_wt(w + d, acc);
    if (d == 0) { st_wt(w + HD, M); st_wt(w + HD + 1, lsum); }
  }
  DBG_T(7);
  DBG_W(13, false);
  if (p.merge && bx == p.n_split - 1) {                     // in-launch merge by the highest split of the (b, head)
    __shared__ float merge_scratch[SPLIT_MERGE_SCRATCH(256)];
    split_merge<256>(p.ws + ((size_t)b * p.nh + head) * p.n_split * 16 * (HD + 2), p.out + (size_t)b * p.L * (p.nh * HD) + head * HD,
                     (size_t)p.nh * HD, p.L, p.n_split, nullptr, merge_scratch);
  }
  DBG_W(11, false);
}

// ---- 128-key tiles (round 2).  Same structure as attn_decode_body, twice the keys per workgroup: wave w owns keys
// [32w, 32w + 32) -- two 16-key S^T blocks (6 MFMA 16x16x32), an 8-value-per-lane softmax, and O^T += V^T.P^T as 6 MFMA
// 16x16x32 whose k index runs over both blocks (k = (j / 4) * 16 + 4g + j % 4, the same permutation on the V^T fragment:
// two 8-byte reads) with P straight from the two accumulators.  Why: at the bench context the 64-key plan needs 1344-1408
// workgroups of which 1280 are resident (5 per CU) -- a second residency round of 64-128 workgroups starts 6 us into the
// launch, and the last head's merge reads 42-44 partials.  With 128-key tiles all 672-704 workgroups are resident at once
// (48 KiB of tile + 3.6 KiB: 3 per CU), every DMA of the launch is in flight from the start, and the merge reads half as
// many partials.  LDS image: [K slice of wave 0..3: 32 keys x 192 B, chunk c of row r at c ^ ((r >> 2) & 3)]
// [V^T tile: 96 rows x 256 B, chunk c (of 16) of row d at c ^ (d & 15): the 16 rows x 2 key groups of a ds_read_b64 lane
// group hit 32 different 8-byte slots].  The merge scratch aliases the (dead) tile region.
// Schedule (the "V2" of profiles/ and DESIGN.md; V1, the same kernel without it, is in the history): the per-workgroup timeline (-DP3V_ATTN_TIMING,
// tools/attn_timeline.py) showed that everything after the DMA issue ran AFTER the whole tile had landed: memory
// returns in order, and the compiler answers a pending `global_load_lds` with vmcnt(0) at every later wait.  V2 issues the
// tile as `buffer_load ... lds` (counted waits stay possible), AFTER the small loads of the new rows, and hides from the
// compiler the LDS accesses that do not depend on the tile (inline asm): Q is rotated, published and fetched while the tile
// is in flight, S^T and the softmax start when the wave's own K slice (its six oldest DMAs, vmcnt(6)) is there and run
// while the V^T tile is still landing.  10.9 -> 10.6 us isolated, 11.1 -> 10.65 us in the decode step.
// FO (k_attn_decode128_o; B = L = 1, every workgroup of the launch resident at once): the layer's o_proj + residual rides in the
// same launch.  The first o_n / 8 non-merging workgroups each own four row PAIRS of W_o (one per wave): once their attention tile
// is consumed they request those rows (12 x 16 bytes per lane, non-temporal -- HBM is otherwise idle while the splits are merged),
// store their partial, then poll the merged attention output words (all-ones sentinel, as the split merge polls its partials; the
// merging workgroups publish them with write-through stores) and finish  x += bf16(W_o . o)  with EXACTLY k_gemv3<1, 1, 6>'s
// arithmetic (lane l holds chunks l, l + 64, ..; dot8 in chunk order; DPP wave sum; resid + bf16 round), so the step's results do
// not depend on whether the projection was fused.  One launch (~5 us) and its boundary leave the layer.  The attention output
// lives in two buffers used by alternate layers: a launch re-arms the one it does not use.
// Round 6: WHERE the roles of the fused launch sit.  Measured (tools/scratch/wg_census_r6.hip): workgroup L of a launch whose
// workgroups are all resident runs on XCD L % 8 and shares its CU with workgroups L +- 256 -- the dispatcher deals them round
// robin over the 256 CUs -- so "virtual CU" v = L % 256, slot = L / 256 says who is co-resident with whom (for speed only: nothing
// below is needed for correctness).  With the (split, head) grid of rounds 4-5 the 32 merging workgroups shared their CUs with
// workgroups streaming 48 KB of W_o each, whose loads sit in the same per-CU memory queue as the merger's polls (a hand-over costs
// 1.1 us into a quiet CU, 2.5-2.9 us into a streaming one: MI355X_MICROARCH.md, handoff-1to1), and W_o requests of a CU's first
// workgroup got ahead of its last workgroup's tile DMA.  Now: G = heads x splits workgroups as a 1-D grid; CUs v < r = G % 256
// carry G / 256 + 1 workgroups, the others G / 256; the mergers are the LAST workgroup of CUs 224 .. 255 (always light CUs: r is
// a multiple of 32 <= 224 with 32 heads), nobody on those CUs streams W_o; the 384 projection units (4 row pairs each) go to the
// last workgroup of every other CU (224 units: their tile is the last of the CU to land, so their W_o requests never precede a
// tile); the remaining 160 units ride on the same workgroups of the least loaded CUs as a second row pair per wave (two_on_last)
// or, the first form of this round, on the second-to-last workgroups (attn_fo_map = 1: their requests precede the CU's last tile
// unless held back ~7.75 us after entry, a gate that would have to follow the cache length; profiles/r06_attn_oproj_placement.txt).
struct FoMap { int bx, by, o_unit, o_unit2; };
__host__ __device__ __forceinline__ bool fo_map_ok(int n_split, int nh, int o_n) { return nh == 32 && o_n == 3072 && n_split >= 13 && n_split * nh <= 768; }
__host__ __device__ __forceinline__ FoMap fo_map(int L, int n_split, int nh, int n_units, bool two_on_last) {
  const int G = n_split * nh, base = G >> 8, r = G & 255;
  const int v = L & 255, slot = L >> 8, ns = base + (v < r ? 1 : 0);
  const int m0 = (base - 1) * 256 + 224;                       // the mergers: ids m0 .. m0 + 31
  FoMap m;
  m.o_unit = m.o_unit2 = -1;
  if (L >= m0 && L < m0 + 32) { m.by = L - m0; m.bx = n_split - 1; return m; }
  const int before = L - m0 < 0 ? 0 : (L - m0 > 32 ? 32 : L - m0);
  const int u = L - before;                                    // rank among the non-merging workgroups: (head, split) head-major
  m.by = u / (n_split - 1);
  m.bx = u - m.by * (n_split - 1);
  if (v < 224) {
    const int from_last = ns - 1 - slot;
    const int idx = base >= 2 || two_on_last ? (v >= r ? v - r : (224 - r) + v) : v;   // second units: light CUs first (one workgroup per light CU and
    const int second = 224 + idx < n_units ? 224 + idx : -1;                             // second-to-last placement: only heavy CUs have a second workgroup)
    if (from_last == 0) { m.o_unit = v; if (two_on_last) m.o_unit2 = second; }
    else if (from_last == 1 && !two_on_last) m.o_unit = second;
  }
  return m;
}

#ifndef P3V_FO_SLEEP
#define P3V_FO_SLEEP 4
#endif
// The projection half of the fused decode launches (k_attn_decode128_o, k_attn_decode128_q8_o): workgroup `o_wl` of the launch's
// non-merging workgroups owns rows [8 o_wl, 8 o_wl + 8) of W_o, one row pair per wave, requested when the workgroup's partial has
// been stored.  KIND: FO_BF16 (k_gemv3's arithmetic), FO_F8 (e4m3 weights, one fp32 scale per row: k_gemv3_f8's) or FO_Q4 (MLX
// 4-bit group-64 in the device layout of p3v_gemv_q4.hip, `o_scale` = its SB words: k_gemv3_q4's).  `lds`: >= 7 KB that nobody
// else uses any more.
enum { FO_NONE = 0, FO_BF16 = 1, FO_F8 = 2, FO_Q4 = 3 };
struct FoP { const void* o_w; const void* o_scale; bf16_t* o_x; const bf16_t* out; int nh; };
template <int KIND, int NU = 1>                                // NU = 2: the workgroup may carry a SECOND unit (o_wl2 >= 0): two row pairs per wave
__device__ __forceinline__ void fo_project(const FoP f, const int o_wl, unsigned char* lds, const int o_wl2 = -1) {
  constexpr bool F8 = KIND == FO_F8, Q4 = KIND == FO_Q4;
  constexpr int HD = 96, NJ = KIND == FO_BF16 ? 6 : 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kd = f.nh * HD;
  const size_t row_bytes = (size_t)Kd * (Q4 ? 1 : F8 ? 2 : 4) / 2;
  const bool two = NU > 1 && o_wl2 >= 0;                       // workgroup-uniform
  int o_u[NU];                                                 // row pair (2 o_u, 2 o_u + 1) of unit k
  u32x4_t ow[NU][2][Q4 ? 1 : NJ];                              // bf16 / e4m3: 16-byte chunks
  u32x2_t oq[NU][2][Q4 ? 3 : 1];                               // 4-bit: 16-weight pieces (8 bytes) + their scale | bias words
  uint32_t osb[NU][2][Q4 ? 3 : 1];
  uint32_t ores[NU];
  float osc[NU][2];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    o_u[k] = (k == 0 ? o_wl : o_wl2) * 4 + wave;
    if (k == 1 && !two) continue;
    const unsigned char* r0p = (const unsigned char*)f.o_w + (size_t)(2 * o_u[k]) * row_bytes;
    if (Q4) {
      const uint32_t* s0 = (const uint32_t*)f.o_scale + (size_t)(2 * o_u[k]) * (Kd / 64) + (lane >> 2);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        oq[k][0][j] = __builtin_nontemporal_load((const u32x2_t*)r0p + j * 64 + lane);
        oq[k][1][j] = __builtin_nontemporal_load((const u32x2_t*)(r0p + row_bytes) + j * 64 + lane);
        osb[k][0][j] = s0[j * 16];
        osb[k][1][j] = s0[Kd / 64 + j * 16];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ow[k][0][j] = __builtin_nontemporal_load((const u32x4_t*)r0p + j * 64 + lane);
        ow[k][1][j] = __builtin_nontemporal_load((const u32x4_t*)(r0p + row_bytes) + j * 64 + lane);
      }
    }
    ores[k] = *(const uint32_t*)(f.o_x + 2 * o_u[k]);
    osc[k][0] = F8 ? ((const float*)f.o_scale)[2 * o_u[k]] : 1.f;
    osc[k][1] = F8 ? ((const float*)f.o_scale)[2 * o_u[k] + 1] : 1.f;
  }
  {
    // Wait for the merged attention output.  1536 waves polling all 1536 words would flood the memory side with uncached
    // 4-byte loads and starve the very workgroups that produce them (measured: every wave timed out); so ONE wave per
    // workgroup watches one word per head (the last word each merging workgroup stores) and sleeps between looks, the others
    // wait at the barrier; then the four waves fetch a quarter of the vector each into LDS (every wave fetching all of it: 18 MB
    // of uncached reads out of one 6 KB region, +2 us), re-fetching the rare straggler word.
    __shared__ int o_timeout;
    const uint32_t* src = (const uint32_t*)f.out;
    if (wave == 0) {
      // bounded by WALL TIME (100 MHz counter), 5 ms: the launch itself lasts ~15 us, and a run whose workgroups are not all resident
      // (a profiler that masks CUs or serialises workgroups, a co-tenant process) must fail FAST and loudly -- NaN row, negative token,
      // api.greedy_loop re-plans with separate launches -- not after 65536 polls per workgroup (rounds 4-5: ~65 ms each, 32 launches per
      // step: whole-graph rocprofv3 --pmc passes did not finish).  P3V_PROFILING=1 plans the separate launches from the start.
      bool to = false;
      const long long t_wait = wall_clock64();
      for (unsigned tries = 0;; ++tries) {
        const uint32_t c = lane < f.nh ? __hip_atomic_load(src + lane * (HD / 2) + HD / 2 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (!__any((c & 0xffffu) == 0xffffu || (c >> 16) == 0xffffu)) break;
        if ((tries & 15u) == 15u && wall_clock64() - t_wait > 500000) { to = true; break; }
        __builtin_amdgcn_s_sleep(P3V_FO_SLEEP);
      }
      if (lane == 0) o_timeout = to;
    }
    TMARK(12);
    __syncthreads();
    // every wave fetches a quarter of the vector (6 words per lane) into LDS, re-fetching the rare straggler word
    uint32_t* xs = (uint32_t*)lds;                              // the K/V tiles are dead by now (barrier above)
    bool timeout = o_timeout != 0;
    {
      const int NW = f.nh * HD / 2 / 4;                        // words per wave: 384
      uint32_t xv[6];
      for (unsigned tries = 0; !timeout; ++tries) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 6; ++j) xv[j] = __hip_atomic_load(src + wave * NW + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 6; ++j) bad |= (xv[j] & 0xffffu) == 0xffffu || (xv[j] >> 16) == 0xffffu;
        if (!__any(bad)) break;
        if (tries >= (1u << 12)) timeout = true;
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) xs[wave * NW + j * 64 + lane] = xv[j];
      if (timeout && lane == 0) o_timeout = 1;
    }
    __syncthreads();
    timeout = o_timeout != 0;
    float* xsum = (float*)(lds + 6144);
    if (Q4) {                                                  // k_gemv3_q4<1, 3>: X = sum of a piece's 16 activations
      if (tid < 192) {
        const u32x4_t a = ((const u32x4_t*)xs)[2 * tid], b = ((const u32x4_t*)xs)[2 * tid + 1];
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) t += (bf16lo(a[j]) + bf16hi(a[j])) + (bf16lo(b[j]) + bf16hi(b[j]));
        xsum[tid] = t;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      if (k == 1 && !two) continue;
      float a0 = 0.f, a1 = 0.f;
      if (Q4) {                                                // piece pc = 64 j + lane
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int pc = j * 64 + lane;
          const u32x4_t xa = ((const u32x4_t*)xs)[2 * pc], xb = ((const u32x4_t*)xs)[2 * pc + 1];
          const float X = xsum[pc], X128 = 128.f * X;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float D = dot8_q4(oq[k][h][j][1], xb, dot8_q4(oq[k][h][j][0], xa, 0.f));
            const float c = bf16lo(osb[k][h][j]) * (D - X128) + bf16hi(osb[k][h][j]) * X;
            if (h == 0) a0 += c; else a1 += c;
          }
        }
        if (k == 0) TMARK(13);
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
      } else if (F8) {                                         // k_gemv3_f8<1, 3>: weight chunk c = 64 j + lane meets x chunks 2c, 2c + 1
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const u32x4_t xa = ((const u32x4_t*)xs)[2 * (j * 64 + lane)], xb = ((const u32x4_t*)xs)[2 * (j * 64 + lane) + 1];
          a0 = dot16_f8(ow[k][0][j], xa, xb, a0);
          a1 = dot16_f8(ow[k][1][j], xa, xb, a1);
        }
        if (k == 0) TMARK(13);
        a0 = wave_sum(a0) * osc[k][0];
        a1 = wave_sum(a1) * osc[k][1];
      } else {                                                 // k_gemv3<1, 1, 6>: chunk 64 j + lane
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const u32x4_t xa = ((const u32x4_t*)xs)[j * 64 + lane];
          a0 = dot8(ow[k][0][j], xa, a0);
          a1 = dot8(ow[k][1][j], xa, a1);
        }
        if (k == 0) TMARK(13);
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
      }
      if (lane == 0) {
        const float v0 = bf16lo(ores[k]) + bf16_round(a0), v1 = bf16hi(ores[k]) + bf16_round(a1);
        *(uint32_t*)(f.o_x + 2 * o_u[k]) = timeout ? 0x7fc07fc0u : pack_bf16x2(v0, v1);     // NaN: loud (api._rows raises)
      }
    }
    TMARK(14);
  }
}

template <int FO = 0>                                          // FO_NONE, FO_BF16 or FO_Q4: the o_proj weights' format
__device__ __forceinline__ void attn_decode_body128(const AttnDecP& p, const int bx, const int by, const int bz, unsigned char* KV,
                                                    const int o_unit = -2, const int o_unit2 = -1) {   // -2: projection units by (head, split) rank (rounds 4-5)
  constexpr int TK = 128, WK = 32, HD = 96, KROW = HD * 2, VROWB = TK * 2, NKS = 3, NDT = 6, CPR = 12;
  constexpr int KS_BYTES = WK * KROW;                          // 6 KiB per wave = 16 x 96 fp32: the wave's O partial parks here
  static_assert(KS_BYTES >= 16 * HD * 4, "a wave's O partial reuses its K slice");
  __shared__ __attribute__((aligned(16))) unsigned char Qs[16 * KROW];
  __shared__ float Ml[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = bz, head = by, kvh = (head * p.grp_magic) >> 16;
  const bool kv_writer = head == kvh * p.grp;
  unsigned char* kslice = KV + wave * KS_BYTES;
  unsigned char* vtile = KV + 4 * KS_BYTES;

  TMARK(0);
#ifdef P3V_ATTN_TIMING
  {                                                            // physical CU of this workgroup (is `L % 256` the co-residency class?)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    TVAL(15, (long long)(((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)));
  }
#endif
  int past = p.past, pad = 0;
  // With d_past, p.past is a LOWER BOUND of the cache length (the prompt length when the graph was captured; negative = no
  // bound known: every tile is requested at once, as in round 3): a
  // tile that starts below it is certainly live and is requested at once; one at or beyond it may lie wholly past the live
  // keys (the cache CAPACITY is prompt + max_tokens, rounded up to the tile) -- it waits for the length and fetches nothing
  // when dead.  (Round 3 fetched every tile of the capacity: 1.06x the live bytes at the bench shape, 1.2x with max_tokens = 512.)
  const int past_lb = p.past;
  if (p.d_past) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(past) : "s"(p.d_past) : "memory");
  if (p.pad_len) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(pad) : "s"(p.pad_len + b) : "memory");

  // ---- tile DMA (static key range: requested before the cache length has arrived)
  const unsigned char* kc = (const unsigned char*)(p.k_cache + ((size_t)b * p.nkv + kvh) * (size_t)p.cache_t * HD);
  bf16_t* vc = p.v_cache + ((size_t)b * p.nkv + kvh) * (size_t)HD * p.cache_t;          // V^T: [hd][cache_t]
  const int kv_lo = bx * TK, kv_hi = min(p.cache_t, kv_lo + TK);
  const size_t vrow = (size_t)p.cache_t * 2;
  auto issue_dma = [&]() {
    const int kv0 = min(kv_lo, p.cache_t - TK);
    const unsigned char* ksrc = kc + (size_t)(kv0 + WK * wave) * KROW;
    const unsigned char* vs = (const unsigned char*)vc + (size_t)kv0 * 2;
    // MUBUF form.  The compiler's wait-count model treats a pending `global_load_lds` as a FLAT access of both memories
    // and turns every later vmcnt wait into vmcnt(0); a pending `buffer_load ... lds` keeps counted waits possible.
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)ksrc, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vs, 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < 6; ++j) {                              // K: LDS slot i = j*64 + lane holds (row i/12, physical chunk i%12)
      const int i = j * 64 + lane, r0 = i / CPR, pc = i - r0 * CPR;
      const unsigned koff = r0 * KROW + ((pc ^ ((r0 >> 2) & 3)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (dec_lptr_t)(kslice + j * 1024), 16, koff, 0, 0, P3V_ATTN_AUX);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {                              // V^T: wave w brings rows 24w..24w+23, four whole 256-B rows per instruction
      const int d = 24 * wave + 4 * j + (lane >> 4);
      const unsigned voff = (unsigned)d * (unsigned)vrow + ((((unsigned)lane & 15) ^ ((unsigned)d & 15)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (dec_lptr_t)(vtile + wave * 6144 + j * 1024), 16, voff, 0, 0, P3V_ATTN_AUX);
    }
  };

  // ---- the L new rows (as attn_decode_body): rotated Q -> LDS, rotated K / V kept until the tile that holds them has landed
  const int row_w = (p.nh + 2 * p.nkv) * HD;
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);
  const int tr = tid / CPR, tc = tid - tr * CPR;
  const bool rtask = tr < p.L;
  RopeRaw qraw, kraw;
  if (rtask) {
    const bf16_t* row = p.qkv + ((size_t)b * p.L + tr) * row_w;
    qraw = rope_fetch(row + head * HD, tc, cos_b + tr * (HD / 2), sin_b + tr * (HD / 2));
    const bf16_t* krow = row + (p.nh + kvh) * HD;
    kraw = qraw;
    kraw.x0 = *(const u32x4_t*)(krow + tc * 8);
    kraw.x1 = *(const u32x4_t*)(krow + (tc < 6 ? tc * 8 + 48 : tc * 8 - 48));
  }
  const bf16_t* vnew = p.qkv + (size_t)b * p.L * row_w + (p.nh + p.nkv + kvh) * HD;   // + r * row_w + d
  const int n_vnew = p.L * HD;
  bf16_t v_early = 0;
  if (tid < n_vnew) {
    const int r = tid / HD;
    v_early = vnew[(size_t)r * row_w + (tid - r * HD)];
  }
  const bool surely_live = p.d_past ? (past_lb < 0 || bx * TK < past_lb) : bx * TK < past_lb + p.L;   // no d_past: the length is exact
  if (surely_live) issue_dma();                                // after the (small) loads above: memory returns in order
  TMARK(7);
  const unsigned qs_lds = (unsigned)(size_t)(dec_lptr_t)Qs;
  if (tid < 16 * CPR) {
    u32x4_t v = {0, 0, 0, 0};
    if (rtask) v = rope_apply(qraw, tc);
    // (an LDS access the compiler must not see: it would wait for the whole tile first)
    asm volatile("ds_write_b128 %0, %1" ::"v"(qs_lds + tr * KROW + ((tc ^ ((tr >> 2) & 3)) << 4)), "v"(v) : "memory");
  }

  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(past), "+s"(pad)::"memory");
  const int total = past + p.L;
  if (!surely_live && bx * TK < total) issue_dma();            // (the waits below count from the most recent requests: unchanged)
  const int kv_end = min(total, kv_hi);
  const int qpos = past + qi;
  const bool qvalid = qi < p.L;
  const float sc2 = p.scale * 1.4426950408889634f;

  const unsigned k_rd = qi * KROW + ((g ^ ((qi >> 2) & 3)) << 4);                                   // + kb*16*KROW + ks*64
  // V^T fragment of key block kb: row 16*dt + qi, keys 32w + 16kb + 4g .. +3 = logical chunk 4w + 2kb + (g >> 1), half g & 1
  unsigned v_rd[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) v_rd[kb] = qi * VROWB + (((4 * wave + 2 * kb + (g >> 1)) ^ qi) << 4) + (g & 1) * 8;   // + dt*16*VROWB

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int kv0 = kv_lo;
  TMARK(8);
  if (kv0 < kv_end) {
    bf16x8_t qf[NKS];
    const bool has_new = kv0 + TK > past;                      // workgroup-uniform: the tile holds new positions
    {                                                          // Q is published and fetched while the tile is in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      u32x4_t q0, q1, q2;
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:64\n\tds_read_b128 %2, %3 offset:128\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(qs_lds + k_rd) : "memory");
      qf[0] = __builtin_bit_cast(bf16x8_t, q0); qf[1] = __builtin_bit_cast(bf16x8_t, q1); qf[2] = __builtin_bit_cast(bf16x8_t, q2);
      if (has_new) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      TMARK(1);
    }
    TMARK(9);
    if (has_new) {
      if (rtask) {
        const int t = past + tr, rr = t - kv0;
        if (rr >= 0 && rr < TK && t < kv_end) {
          const u32x4_t kn = rope_apply(kraw, tc);
          const int row = rr & 31;
          *(u32x4_t*)(KV + (rr >> 5) * KS_BYTES + row * KROW + ((tc ^ ((row >> 2) & 3)) << 4)) = kn;
          if (kv_writer) *(u32x4_t*)(p.k_cache + (((size_t)b * p.nkv + kvh) * p.cache_t + t) * HD + tc * 8) = kn;   // phi.py:545
        }
      }
#pragma unroll 1
      for (int idx = tid; idx < n_vnew; idx += 256) {
        const int r = idx / HD, d = idx - r * HD, t = past + r, rr = t - kv0;
        if (rr >= 0 && rr < TK && t < kv_end) {
          const bf16_t val = idx == tid ? v_early : vnew[(size_t)r * row_w + d];
          *(bf16_t*)(vtile + d * VROWB + (((rr >> 3) ^ (d & 15)) << 4) + (rr & 7) * 2) = val;
          if (kv_writer) vc[(size_t)d * p.cache_t + t] = val;                                               // phi.py:546
        }
      }
      __syncthreads();
    }

    f32x4_t s[2];
    if (!has_new) {
      // this wave's K slice = its six oldest DMAs: S^T and the softmax run while the V^T tile is still landing
      u32x4_t kf[2][NKS];
      const unsigned ka = (unsigned)(size_t)(dec_lptr_t)kslice + k_rd;
      asm volatile("s_waitcnt vmcnt(6)\n\t"
                   "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:64\n\tds_read_b128 %2, %6 offset:128\n\t"
                   "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:3136\n\tds_read_b128 %5, %6 offset:3200\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(kf[0][0]), "=&v"(kf[0][1]), "=&v"(kf[0][2]), "=&v"(kf[1][0]), "=&v"(kf[1][1]), "=&v"(kf[1][2])
                   : "v"(ka) : "memory");
      static_assert(16 * KROW == 3072, "second key block of the wave's slice");
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[kb][ks]), qf[ks], s[kb], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8_t kf = *(const bf16x8_t*)(kslice + kb * 16 * KROW + k_rd + ks * 64);
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
      }
    }
    float m_t = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + WK * wave + 16 * kb + 4 * g + r;
        const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
        s[kb][r] = vis ? s[kb][r] * sc2 : -INFINITY;
        m_t = fmaxf(m_t, s[kb][r]);
      }
    m_t = rows_max(m_t);
    const float m_use = m_t == -INFINITY ? 0.f : m_t;
    float l_t = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - m_use);
        l_t += s[kb][r];
      }
    l_run = rows_sum(l_t);
    m_run = m_t;
    if (l_run == 12345.f) TVAL(15, 1);                         // consume the softmax before the mark
    TMARK(10);
    const u32x4_t pw = {pack_bf16x2(s[0][0], s[0][1]), pack_bf16x2(s[0][2], s[0][3]), pack_bf16x2(s[1][0], s[1][1]), pack_bf16x2(s[1][2], s[1][3])};
    const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
    if (!has_new) {                                            // the V^T tile is every wave's DMA: all landed, then the barrier
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      const u32x2_t a0 = *(const u32x2_t*)(vtile + v_rd[0] + d * 16 * VROWB), a1 = *(const u32x2_t*)(vtile + v_rd[1] + d * 16 * VROWB);
      const u32x4_t aw = {a0[0], a0[1], a1[0], a1[1]};
      o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aw), pf, o[d], 0, 0, 0);
    }
  }

  // ---- merge the four wave partials through the (dead) K slices, one (m, l, O) partial per workgroup to `ws`
  const bool merger = p.merge && bx == p.n_split - 1;       // the highest split of a (b, head) merges in-launch
  float* own = (float*)(vtile + 4096);                         // its partial: [16][HD + 2] in the dead V^T tile (merge scratch: first 4 KiB)
  static_assert(SPLIT_MERGE_SCRATCH(256) * 4 <= 4096 && 4096 + 16 * (HD + 2) * 4 <= 96 * 256, "merge scratch + own partial fit the V^T tile");
  if (o[0][0] == 12345.f) TVAL(15, 2);
  TMARK(11);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  TMARK(2);
  // fused o_proj: this wave's row pair is requested once the partial is stored (below): requested here, 18.9 MB of weight
  // loads compete with the other workgroups' K/V tiles and the whole attention phase runs 0.6 us later (tools/attn_o_timeline.py)
  const int o_wl = o_unit != -2 ? o_unit : by * (p.n_split - 1) + bx;      // projection unit: rows [8 o_wl, 8 o_wl + 8)
  const bool o_worker = FO && !merger && bx < p.n_split - 1 && o_wl >= 0 && o_wl * 8 < p.o_n;
  if (FO) {
    if (bx == 0 && by == 0) {                                  // re-arm the other layer parity's buffer (nobody reads it in this launch)
      uint32_t* ra = (uint32_t*)p.o_rearm;
      for (int i = tid; i < p.nh * HD / 2; i += 256) ra[i] = 0xffffffffu;
    }
  }
  if (qvalid) {
    float* Ow = (float*)kslice + qi * HD;
#pragma unroll
    for (int d = 0; d < NDT; ++d) *(f32x4_t*)(Ow + 16 * d + 4 * g) = o[d];
    if (g == 0) { Ml[wave][qi][0] = m_run; Ml[wave][qi][1] = l_run; }
  }
  __syncthreads();
#pragma unroll 1
  for (int idx = tid; idx < n_vnew; idx += 256) {
    const int q = idx / HD, d = idx - q * HD;
    float mk[4], M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mk[k] = Ml[k][q][0]; M = fmaxf(M, mk[k]); }
    const float Mu = M == -INFINITY ? 0.f : M;
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = __builtin_amdgcn_exp2f(mk[k] - Mu);
      acc += c * ((const float*)(KV + k * KS_BYTES))[q * HD + d];
      lsum += c * Ml[k][q][1];
    }
    if (merger) {                                              // stays in LDS: split_merge takes it from there
      own[q * (HD + 2) + d] = acc;
      if (d == 0) { own[q * (HD + 2) + HD] = M; own[q * (HD + 2) + HD + 1] = lsum; }
    } else {
      float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + bx) * 16 + q) * (HD + 2);
      st_wt(w + d, acc);
      if (d == 0) { st_wt(w + HD, M); st_wt(w + HD + 1, lsum); }
    }
  }
  TMARK(3);
  if (merger)
    split_merge<256>(p.ws + ((size_t)b * p.nh + head) * p.n_split * 16 * (HD + 2), p.out + (size_t)b * p.L * (p.nh * HD) + head * HD,
                     (size_t)p.nh * HD, p.L, p.n_split, own, (float*)vtile, FO != 0);
  if (FO && o_worker) {
    fo_project<FO, 2>(FoP{p.o_w, p.o_sb, p.o_x, p.out, p.nh}, o_wl, KV, o_unit2);
  }
}

__global__ void __launch_bounds__(256) k_attn_decode128(AttnDecP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char KV[4 * 6144 + 96 * 256];   // [K slice x 4 | V^T tile] = 48 KiB
  attn_decode_body128<FO_NONE>(p, blockIdx.x, blockIdx.y, blockIdx.z, KV);
}

__global__ void __launch_bounds__(256) k_attn_decode128_o(AttnDecP p) {            // + the layer's o_proj + residual (body128<true>)
  __shared__ __attribute__((aligned(1024))) unsigned char KV[4 * 6144 + 96 * 256];
  int bx = blockIdx.x, by = blockIdx.y, o_unit = -2, o_unit2 = -1;
  if (p.fo_remap) {                                            // 1-D grid, roles placed by virtual CU (fo_map)
    const FoMap m = fo_map((int)blockIdx.x, p.n_split, p.nh, p.o_n / 8, p.fo_remap == 2);
    bx = m.bx, by = m.by, o_unit = m.o_unit, o_unit2 = m.o_unit2;
  }
  attn_decode_body128<FO_BF16>(p, bx, by, blockIdx.z, KV, o_unit, o_unit2);
}

__global__ void __launch_bounds__(256) k_attn_decode128_o4(AttnDecP p) {           // the same on MLX 4-bit group-64 o_proj weights
  __shared__ __attribute__((aligned(1024))) unsigned char KV[4 * 6144 + 96 * 256];
  int bx = blockIdx.x, by = blockIdx.y, o_unit = -2, o_unit2 = -1;
  if (p.fo_remap) {                                            // 1-D grid, roles placed by virtual CU (fo_map)
    const FoMap m = fo_map((int)blockIdx.x, p.n_split, p.nh, p.o_n / 8, p.fo_remap == 2);
    bx = m.bx, by = m.by, o_unit = m.o_unit, o_unit2 = m.o_unit2;
  }
  attn_decode_body128<FO_Q4>(p, bx, by, blockIdx.z, KV, o_unit, o_unit2);
}

__global__ void __launch_bounds__(256) k_attn_decode(AttnDecP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char KV[4 * 6144];   // [wave]{K slice | V^T slice}
  attn_decode_body(p, blockIdx.x, blockIdx.y, blockIdx.z, KV);
}

template <int TK>
__global__ void __launch_bounds__(64) k_attn_decode_stream(AttnDecP p) {
  constexpr int HD = 96, KSTR = HD * 2 + 16, VSTR = TK * 2 + 16, NKS = 3, NDT = 6, CPR = 12;
  constexpr int KIT = TK * CPR / 64, VCH = TK / 8, VIT = HD * VCH / 64;   // 16-byte loads per lane: K, V^T
  __shared__ __attribute__((aligned(16))) unsigned char Ks[TK * KSTR];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[HD * VSTR];
  const int lane = threadIdx.x, g = lane >> 4, qi = lane & 15;
  const int b = blockIdx.z, head = blockIdx.y, kvh = (head * p.grp_magic) >> 16;
  const bool kv_writer = head == kvh * p.grp;
  const float sc2 = p.scale * 1.4426950408889634f;            // scale * log2(e): softmax runs on exp2
  const int row_w = (p.nh + 2 * p.nkv) * HD;                  // qkv row width
  const bf16_t* kc = p.k_cache + ((size_t)b * p.nkv + kvh) * (size_t)p.cache_t * HD;
  bf16_t* vc = p.v_cache + ((size_t)b * p.nkv + kvh) * (size_t)HD * p.cache_t;          // V^T: [hd][cache_t]
  // The key range of a split is STATIC (a function of the cache capacity, not of the current length):
  // its first tile can be requested before the cache length `past` has even arrived from HBM.
  const int chunk = p.chunk;
  const int kv_lo = blockIdx.x * chunk, kv_hi = min(p.cache_t, kv_lo + chunk);

  // ---- tile registers: K TK rows x 12 chunks, V^T 96 rows x TK/8 chunks, all requested before anything is
  //      used.  Rows beyond the live length are read too (allocated, finite V^T / masked K) and ignored.
  u32x4_t kreg[KIT], vreg[VIT];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int i = it * 64 + lane;
      kreg[it] = __builtin_nontemporal_load((const u32x4_t*)(kc + (size_t)(kv0 + i / CPR) * HD + (i % CPR) * 8));
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int i = it * 64 + lane;
      vreg[it] = __builtin_nontemporal_load((const u32x4_t*)(vc + (size_t)(i / VCH) * p.cache_t + kv0 + (i % VCH) * 8));
    }
  };
  if (kv_lo < kv_hi) load_tile(kv_lo);                         // cache_t % TK == 0: the tile is always in bounds

  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b] : 0;
  int kv_begin = kv_lo;
  const int kv_end = min(total, kv_hi);
  if (pad > kv_begin) kv_begin = pad & ~(TK - 1);
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);

  // positions [past, total) that fall in this tile: K rotated from the qkv row, V copied -- written straight into
  // the LDS tile (after the bulk register->LDS store) and appended to the cache by the writer block.  Rolled
  // loops on purpose: this is the rare path (one tile per head) and must not cost registers.
  auto patch_new = [&](int kv0) {
    const int n0 = max(past, kv0), n1 = min(kv_end, kv0 + TK), n_new = n1 - n0;
#pragma unroll 1
    for (int w = lane; w < n_new * CPR; w += 64) {
      const int t = n0 + w / CPR, c = w % CPR, r = t - past;
      const bf16_t* row = p.qkv + ((size_t)b * p.L + r) * row_w;
      const u32x4_t kn = rope_chunk(row + (p.nh + kvh) * HD, c, cos_b + r * (HD / 2), sin_b + r * (HD / 2));
      *(u32x4_t*)(Ks + (t - kv0) * KSTR + c * 16) = kn;
      if (kv_writer) *(u32x4_t*)(p.k_cache + (((size_t)b * p.nkv + kvh) * p.cache_t + t) * HD + c * 8) = kn;   // phi.py:545
    }
#pragma unroll 1
    for (int w = lane; w < n_new * HD; w += 64) {
      const int t = n0 + w / HD, d = w % HD, r = t - past;
      const bf16_t val = p.qkv[((size_t)b * p.L + r) * row_w + (p.nh + p.nkv + kvh) * HD + d];
      *(bf16_t*)(Vt + d * VSTR + (t - kv0) * 2) = val;
      if (kv_writer) vc[(size_t)d * p.cache_t + t] = val;                                                   // phi.py:546
    }
  };
  if (kv_begin > kv_lo && kv_begin < kv_end) load_tile(kv_begin);   // left padding skipped whole tiles: reload


  const bool qvalid = qi < p.L;
  const int qpos = past + qi;
  bf16x8_t qf[NKS];
  {
    const bf16_t* qrow = p.qkv + ((size_t)b * p.L + (qvalid ? qi : 0)) * row_w + head * HD;
    const float* ct = cos_b + (qvalid ? qi : 0) * (HD / 2);
    const float* st = sin_b + (qvalid ? qi : 0) * (HD / 2);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = rope_chunk(qrow, 4 * ks + g, ct, st);
      if (!qvalid) v = (u32x4_t){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += TK) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int i = it * 64 + lane;
      *(u32x4_t*)(Ks + (i / CPR) * KSTR + (i % CPR) * 16) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int i = it * 64 + lane;
      *(u32x4_t*)(Vt + (i / VCH) * VSTR + (i % VCH) * 16) = vreg[it];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's LDS writes have landed (single-wave block)
    if (kv0 + TK > past) {                                     // wave-uniform: only the tile(s) holding new positions
      patch_new(kv0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (kv0 + TK < kv_end) load_tile(kv0 + TK);                // next tile streams in under the MFMAs below

    f32x4_t s[TK / 16];
#pragma unroll
    for (int st = 0; st < TK / 16; ++st) {
      s[st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + (16 * st + qi) * KSTR + (32 * ks + 8 * g) * 2);
        s[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[st], 0, 0, 0);
      }
    }
    float m_t = -INFINITY;
#pragma unroll
    for (int st = 0; st < TK / 16; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + 16 * st + 4 * g + r;
        const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
        const float v = vis ? s[st][r] * sc2 : -INFINITY;               // log2 domain: scale*log2(e) folded in
        s[st][r] = v;
        m_t = fmaxf(m_t, v);
      }
    m_t = rows_max(m_t);
    const float m_new = fmaxf(m_run, m_t);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float l_t = 0.f;
#pragma unroll
    for (int st = 0; st < TK / 16; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[st][r] - m_use);
        s[st][r] = e;
        l_t += e;
      }
    l_t = rows_sum(l_t);
    l_run = l_run * alpha + l_t;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[d] *= alpha;
#pragma unroll
    for (int st = 0; st < TK / 32; ++st) {
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // LDS reads of this tile precede the next tile's writes
  }
  if (qvalid) {
    float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + blockIdx.x) * 16 + qi) * (HD + 2);
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) st_wt(w + 16 * d + 4 * g + r, o[d][r]);
    if (g == 0) { st_wt(w + HD, m_run); st_wt(w + HD + 1, l_run); }
  }
  if (p.merge && blockIdx.x == p.n_split - 1) {
    __shared__ float merge_scratch[SPLIT_MERGE_SCRATCH(64)];
    split_merge<64>(p.ws + ((size_t)b * p.nh + head) * p.n_split * 16 * (HD + 2), p.out + (size_t)b * p.L * (p.nh * HD) + head * HD,
                    (size_t)p.nh * HD, p.L, p.n_split, nullptr, merge_scratch);
  }
}

// merge split-KV partials: one block of G = 4 or 8 64-lane groups per (b, head, query): thread (grp, d) loads
// (m, l, o[d]) of its share of the splits (batches of 4 independent loads), reduces them against its own
// running max, and the groups are merged through LDS.
// (the body is shared with k_attn_combine_o below: WT = the row goes out as 48 packed words with write-through stores, for
//  consumers polling them in the same launch; the value of every element is computed by the same expression either way)
template <bool WT>
__device__ __forceinline__ void combine_rows(const float* __restrict__ ws, bf16_t* __restrict__ out, int L, int nh, int hd, int n_split, int block) {
  __shared__ float pm[CMB_G], pl[CMB_G];
  __shared__ float part[CMB_G][128];
  const int qi = block % L, head = (block / L) % nh, b = block / (L * nh);
  const float* base = ws + (((size_t)b * nh + head) * n_split * 16 + qi) * (hd + 2);
  const size_t sstr = (size_t)16 * (hd + 2);
  const int t = threadIdx.x, grp = t >> 6, d0 = t & 63;       // G groups x 64 lanes; lane handles d0 and d0+64
  const int G = blockDim.x >> 6;                              // 4, 8 or 16 (combine_threads)
  const int per = (n_split + G - 1) / G, s0 = grp * per, s1 = min(n_split, s0 + per);
  const bool two = d0 + 64 < hd;
  float m = -INFINITY, l = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll 4
  for (int s = s0; s < s1; ++s) {
    const float ms = base[s * sstr + hd], ls = base[s * sstr + hd + 1];
    const float o0 = base[s * sstr + d0], o1 = two ? base[s * sstr + d0 + 64] : 0.f;
    const float mn = fmaxf(m, ms);
    const float mu = mn == -INFINITY ? 0.f : mn;
    const float ca = __builtin_amdgcn_exp2f(m - mu), cb = __builtin_amdgcn_exp2f(ms - mu);   // partial maxima are log2-domain; exp2(-inf) = 0 covers empty partials
    l = l * ca + ls * cb;
    a0 = a0 * ca + o0 * cb;
    a1 = a1 * ca + o1 * cb;
    m = mn;
  }
  {                                                            // `ws` is all-sentinel between launches (see split_merge)
    float* wbase = const_cast<float*>(base);
    const float sentinel = __builtin_bit_cast(float, WS_SENTINEL);
    for (int s = s0; s < s1; ++s) {
      wbase[s * sstr + d0] = sentinel;
      if (two) wbase[s * sstr + d0 + 64] = sentinel;
      if (d0 == 0) { wbase[s * sstr + hd] = sentinel; wbase[s * sstr + hd + 1] = sentinel; }
    }
  }
  if (d0 == 0) { pm[grp] = m; pl[grp] = l; }
  part[grp][d0] = a0;
  if (two) part[grp][d0 + 64] = a1;
  __syncthreads();
  auto elem = [&](int e) {
    float M = pm[0];
    for (int k = 1; k < G; ++k) M = fmaxf(M, pm[k]);
    const float Mu = M == -INFINITY ? 0.f : M;
    float acc = 0.f, lsum = 0.f;
#pragma unroll 4
    for (int k = 0; k < G; ++k) {
      const float c = __builtin_amdgcn_exp2f(pm[k] - Mu);
      acc += c * part[k][e];
      lsum += c * pl[k];
    }
    return lsum > 0.f ? acc / lsum : 0.f;
  };
  bf16_t* row = out + ((size_t)b * L + qi) * (size_t)(nh * hd) + head * hd;
  if (WT) {
    if (t < hd / 2) __hip_atomic_store((uint32_t*)row + t, pack_bf16x2(elem(2 * t), elem(2 * t + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (t < hd) {
    row[t] = f32_to_bf16(elem(t));
  }
}

__global__ void __launch_bounds__(64 * CMB_G) k_attn_combine2(const float* __restrict__ ws, bf16_t* __restrict__ out, int L,
                                                              int nh, int hd, int n_split) {
  combine_rows<false>(ws, out, L, nh, hd, n_split, (int)blockIdx.x);
}

// Round 6, long contexts (the split-KV plans whose partials are merged by a launch of their own: more than 48 tiles of 128 keys, i.e.
// every context beyond ~6k tokens up to the model's 128k): the merge launch also carries the layer's o_proj + residual (B = L = 1),
// as k_attn_decode128_o does for the one-tile plans.  Workgroups 0 .. nh - 1 merge one head each (k_attn_combine2's arithmetic, the row
// published as 48 write-through words); the others are fo_project units (two row-pair quartets each): their W_o rows are requested at
// once and the dot products start when the 32 rows have been seen.  224 workgroups: every one resident (one per CU).
// 32k keys: combine 4.9 us + o_proj 5.0 us -> one launch.
template <int KIND>
__global__ void __launch_bounds__(512) k_attn_combine_o(const float* __restrict__ ws, bf16_t* __restrict__ out, bf16_t* __restrict__ o_rearm,
                                                        const void* o_w, const void* o_sb, bf16_t* o_x, int nh, int n_split, int n_units) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[7 * 1024 + 768];
  if ((int)blockIdx.x < nh) {
    if (blockIdx.x == 0) {                                     // re-arm the other layer parity's buffer (nobody reads it in this launch)
      uint32_t* ra = (uint32_t*)o_rearm;
      for (int i = threadIdx.x; i < nh * 96 / 2; i += blockDim.x) ra[i] = 0xffffffffu;
    }
    combine_rows<true>(ws, out, 1, nh, 96, n_split, (int)blockIdx.x);
    return;
  }
  if (threadIdx.x >= 256) return;                              // (the merging workgroups use all 64 x G threads, a unit four waves)
  const int n_uw = (int)gridDim.x - nh, u = (int)blockIdx.x - nh;
  fo_project<KIND, 2>(FoP{o_w, o_sb, o_x, out, nh}, u, lds, u + n_uw < n_units ? u + n_uw : -1);
}

// B = L = 1, 32 x 96: one workgroup per head + one per two row-pair quartets, all resident at once (one per CU, 8 CUs of head-room)
static bool combine_o_ok(int B, int L, int n_heads, int hd, int n_split, int o_n) {
  if (B != 1 || L != 1 || hd != 96 || n_heads * hd != 3072 || o_n <= 0 || o_n % 8) return false;
  if (n_split < 1 || n_split > 128 || combine_threads(n_split) > 512) return false;
  int dev = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
  return n_heads + (o_n / 8 + 1) / 2 <= pr.multiProcessorCount - 8;
}
template <int KIND>
static int launch_combine_o(const float* ws, void* out, void* o_rearm, const void* o_w, const void* o_sb, void* o_x, int n_heads, int n_split,
                            int o_n, hipStream_t s) {
  const int n_units = o_n / 8, grid_o = n_heads + (n_units + 1) / 2;
  hipLaunchKernelGGL(k_attn_combine_o<KIND>, dim3(grid_o), dim3(combine_threads(n_split)), 0, s, ws, (bf16_t*)out, (bf16_t*)o_rearm, o_w, o_sb,
                     (bf16_t*)o_x, n_heads, n_split, n_units);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// cos/sin rows of the L positions [past, past+L) of every batch row -> compact [B, L, half] buffers, so that
// the decode attention of all layers reads them at addresses that do not depend on the device-side cache length
__global__ void k_stage_rope(const float* __restrict__ cos_t, const float* __restrict__ sin_t, const int32_t* d_past,
                             int past, float* __restrict__ cos_o, float* __restrict__ sin_o, int L, int tab_t, int half) {
  if (d_past) past = *d_past;
  const int b = blockIdx.x / L, r = blockIdx.x % L;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    cos_o[((size_t)b * L + r) * half + i] = cos_t[((size_t)b * tab_t + past + r) * half + i];
    sin_o[((size_t)b * L + r) * half + i] = sin_t[((size_t)b * tab_t + past + r) * half + i];
  }
}

extern "C" int p3v_stage_rope(const float* cos_t, const float* sin_t, int past, const int32_t* d_past, float* cos_out,
                              float* sin_out, int B, int L, int tab_t, int half_dim, void* stream) {
  if (!cos_t || !sin_t || !cos_out || !sin_out || B <= 0 || L <= 0) return P3V_ERR_ARG;
  hipLaunchKernelGGL(k_stage_rope, dim3(B * L), dim3(64), 0, (hipStream_t)stream, cos_t, sin_t, d_past, past, cos_out,
                     sin_out, L, tab_t, half_dim);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

#ifdef P3V_ATTN_DEBUG
extern "C" int p3v_debug_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// The fused form needs: one row, one new token, a 3072-wide attention output (k_gemv3<1, 1, 6>'s shape), the 128-key plan with the
// in-launch merge, enough non-merging workgroups for the o_n / 8 row-pair quartets, and EVERY workgroup of the launch resident at
// once (3 per CU) -- the polling workgroups then cannot keep a producer from being scheduled.
extern "C" int p3v_attention_decode_can_fuse_oproj(int B, int L, int n_heads, int hd, int n_split, int cache_t, int o_n, int merge_in_launch) {
  if (B != 1 || L != 1 || hd != 96 || n_heads * hd != 3072 || o_n <= 0 || o_n % 8) return 0;
  if (!merge_in_launch)      // the partials are merged by a launch of their own: that launch takes the o_proj (k_attn_combine_o) -- answer 2
    return combine_o_ok(B, L, n_heads, hd, n_split, o_n) ? 2 : 0;
  if (!(n_split * 128 >= cache_t && cache_t % 128 == 0 && n_split * 64 < cache_t) || n_split < 2) return 0;
  if ((long)(n_split - 1) * n_heads * 8 < o_n) return 0;
  // every workgroup of the launch must be resident at once (the projecting ones wait for the merging ones): ask the runtime how
  // many of THIS kernel fit a CU rather than assume the 3 that its 52 KB of LDS and ~100 VGPRs give today
  // (cached PER DEVICE: ADVICE r04 -- a process-wide value was whatever the first device that asked had)
  static long capacity_of[P3V_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= P3V_MAX_DEVICES) return 0;
  if (!capacity_of[dev]) {
    int per_cu = 0;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
    int per_cu4 = 0;                                           // (the bf16 and the 4-bit form of the launch: the smaller of the two)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_decode128_o, 256, 0) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, k_attn_decode128_o4, 256, 0) != hipSuccess) return 0;
    per_cu = min(per_cu, per_cu4);
    if (per_cu < 1) return 0;
    capacity_of[dev] = (long)per_cu * (pr.multiProcessorCount - 8);   // 8 CUs of head-room: a grid of exactly the queried capacity
  }                                                                   // did not co-reside in tools/scratch/persistent_chain.hip
  const long capacity = capacity_of[dev];
  return (long)B * n_heads * n_split <= capacity;
}

extern "C" int p3v_attention_decode_fused_role(int wg, int n_heads, int n_split, int o_n, int map, int* out4) {
  if (!out4 || wg < 0 || n_split < 2 || n_heads < 1 || wg >= n_heads * n_split || (map != 1 && map != 2)) return P3V_ERR_ARG;
  if (!fo_map_ok(n_split, n_heads, o_n)) return P3V_ERR_UNSUPPORTED;
  const FoMap m = fo_map(wg, n_split, n_heads, o_n / 8, map == 2);
  out4[0] = m.bx, out4[1] = m.by, out4[2] = m.o_unit, out4[3] = m.o_unit2;
  return P3V_OK;
}

extern "C" int p3v_attention_decode(const p3v_attn_decode_args_t* a, void* stream) {
  if (!a || !a->qkv || !a->cos_t || !a->sin_t || !a->k_cache || !a->v_cache || !a->out || !a->ws) return P3V_ERR_ARG;
  if (a->hd != 96) return P3V_ERR_UNSUPPORTED;
  if (a->B <= 0 || a->L <= 0 || a->L > P3V_DECODE_MAX_L || a->n_heads % a->n_kv) return P3V_ERR_ARG;
  if (a->n_split < 1 || a->n_split > 128 || a->cache_t % 64) return P3V_ERR_ARG;
  if (((uintptr_t)a->ws | (uintptr_t)a->out) & 15) return P3V_ERR_ARG;      // 16-byte write-through stores of the partial records / merged rows
  const int grp = a->n_heads / a->n_kv;
  const int chunk = ((a->cache_t + a->n_split - 1) / a->n_split + 63) & ~63;
  AttnDecP p = {a->qkv, a->cos_t, a->sin_t, a->k_cache, a->v_cache, a->pad_len, a->d_past, a->ws,
                a->B, a->L, a->n_heads, a->n_kv, a->past, a->cache_t, a->rope_bstride, a->n_split, a->scale,
                chunk, grp, (65536 + grp - 1) / grp, a->merge_in_launch, (bf16_t*)a->out,
                a->o_proj_w, a->o_proj_sb, (bf16_t*)a->o_proj_x, (bf16_t*)a->o_rearm, a->o_n, 0};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a->n_split, a->n_heads, a->B);
  const int fuse_form = a->o_proj_w ? p3v_attention_decode_can_fuse_oproj(a->B, a->L, a->n_heads, a->hd, a->n_split, a->cache_t, a->o_n, a->merge_in_launch) : 0;
  if (a->o_proj_w && !fuse_form) return P3V_ERR_UNSUPPORTED;
  if (a->o_proj_w && (!a->o_proj_x || !a->o_rearm || ((uintptr_t)a->o_proj_w | (uintptr_t)a->o_proj_x | (uintptr_t)a->o_rearm) & 15)) return P3V_ERR_ARG;
  if (a->o_proj_w && a->o_proj_sb && ((uintptr_t)a->o_proj_sb & 3)) return P3V_ERR_ARG;
  if (fuse_form == 1) {                                        // attention + o_proj + residual in one launch
    if (p3v_tuning().attn_fo_map && fo_map_ok(a->n_split, a->n_heads, a->o_n)) {
      p.fo_remap = p3v_tuning().attn_fo_map;                    // 1: second units on the second-to-last workgroups, 2: on the last ones
      grid = dim3(a->n_split * a->n_heads, 1, 1);
    }
    if (a->o_proj_sb) {                                        // 4-bit group-64 o_proj weights
      if ((uintptr_t)a->o_proj_sb & 3) return P3V_ERR_ARG;
      hipLaunchKernelGGL(k_attn_decode128_o4, grid, dim3(256), 0, s, p);
    } else {
      hipLaunchKernelGGL(k_attn_decode128_o, grid, dim3(256), 0, s, p);
    }
    P3V_CHECK_LAUNCH();
    return P3V_OK;
  }
  if (a->n_split * 64 >= a->cache_t) hipLaunchKernelGGL(k_attn_decode, grid, dim3(256), 0, s, p);
  else if (a->n_split * 128 >= a->cache_t && a->cache_t % 128 == 0) hipLaunchKernelGGL(k_attn_decode128, grid, dim3(256), 0, s, p);   // 128-key tiles
  else hipLaunchKernelGGL(k_attn_decode_stream<64>, grid, dim3(64), 0, s, p);
  P3V_CHECK_LAUNCH();
  if (a->merge_in_launch) return P3V_OK;                       // the last workgroup of every (b, head) merged in-kernel
  if (fuse_form == 2)                                          // the merge launch carries the o_proj + residual
    return a->o_proj_sb ? launch_combine_o<FO_Q4>(a->ws, a->out, a->o_rearm, a->o_proj_w, a->o_proj_sb, a->o_proj_x, a->n_heads, a->n_split, a->o_n, s)
                        : launch_combine_o<FO_BF16>(a->ws, a->out, a->o_rearm, a->o_proj_w, nullptr, a->o_proj_x, a->n_heads, a->n_split, a->o_n, s);
  hipLaunchKernelGGL(k_attn_combine2, dim3(a->B * a->n_heads * a->L), dim3(combine_threads(a->n_split)), 0, s, a->ws, a->out, a->L, a->n_heads,
                     a->hd, a->n_split);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// =====================================================================================
// int8 KV cache (quantize_cache=True; BASELINE config 5; replaces the reference's 4-bit group-32 prompt
// cache, phi.py:528-540): K rows and V^T columns are stored as offset-binary bytes
// u = round(x / s) + 128 with one fp32 scale s per (batch row, kv head, token); halves the bytes a decode
// step streams.  As in the reference, the PREFILL attends over the exact keys/values and only the
// stored copy is quantised (phi.py:531-533); a decode step attends over the dequantised cache plus its
// own exact new row.
//
// p3v_kv_quantize: bf16 K [B,nkv,Ts,hd] / V^T [B,nkv,hd,Ts] rows [t0, t0+n) -> u8 caches + scales.
__global__ void __launch_bounds__(256) k_kv_quantize(const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
                                                     uint8_t* __restrict__ k8, uint8_t* __restrict__ v8,
                                                     float* __restrict__ ksc, float* __restrict__ vsc, int hd, int src_t,
                                                     int dst_t, int t0, int n_tok) {
  __shared__ float vmax[64];
  __shared__ bf16_t vtile[96 * 66];
  const int bh = blockIdx.y, tt = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- K: one wave per token row (4 rows per pass)
  for (int r = wave; r < 64; r += 4) {
    const int t = t0 + tt + r;
    if (tt + r >= n_tok) break;
    const bf16_t* row = k + ((size_t)bh * src_t + t) * hd;
    const float a = lane < hd / 2 ? bf16_to_f32(row[2 * lane]) : 0.f, b = lane < hd / 2 ? bf16_to_f32(row[2 * lane + 1]) : 0.f;
    const float amax = wave_max(fmaxf(fabsf(a), fabsf(b)));
    const float s = amax > 0.f ? amax / 127.f : 1.f, inv = 1.f / s;
    if (lane < hd / 2) {
      const int qa = (int)rintf(a * inv) + 128, qb = (int)rintf(b * inv) + 128;
      *(uint16_t*)(k8 + ((size_t)bh * dst_t + t) * hd + 2 * lane) = (uint16_t)(qa | (qb << 8));
    }
    if (lane == 0) ksc[(size_t)bh * dst_t + t] = s;
  }
  // ---- V^T: tile [hd][64 tokens] through LDS, per-token amax over d
  for (int i = tid; i < hd * 64; i += 256) {
    const int d = i >> 6, c = i & 63;
    vtile[d * 66 + c] = tt + c < n_tok ? vt[((size_t)bh * hd + d) * src_t + t0 + tt + c] : (bf16_t)0;
  }
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
    for (int d = 0; d < hd; ++d) m = fmaxf(m, fabsf(bf16_to_f32(vtile[d * 66 + tid])));
    vmax[tid] = m > 0.f ? m / 127.f : 1.f;
    if (tt + tid < n_tok) vsc[(size_t)bh * dst_t + t0 + tt + tid] = vmax[tid];
  }
  __syncthreads();
  for (int i = tid; i < hd * 64; i += 256) {
    const int d = i >> 6, c = i & 63;
    if (tt + c < n_tok)
      v8[((size_t)bh * hd + d) * dst_t + t0 + tt + c] = (uint8_t)((int)rintf(bf16_to_f32(vtile[d * 66 + c]) / vmax[c]) + 128);
  }
}

// Vector-access version (t0 % 8 == 0, src_t % 8 == 0, dst_t % 8 == 0, hd == 96: every prefill): the first version went
// through 2-byte loads and one dependent row per wave and pass (34 us per layer at 2531 tokens, 1.1 ms of a config-5
// prefill).  K: thread = one 16-byte chunk (8 dims) of a row, 12 chunks per row, 16 lanes per row -> row maximum by DPP
// inside the 16-lane row; V^T: thread = 8 tokens of one dim, per-token maximum over the 96 dims through LDS atomics on
// the (non-negative) float bits.  Same arithmetic as k_kv_quantize: s = amax / 127, code = rint(x * (1 / s)) + 128 for K,
// rint(x / s) + 128 for V.
__global__ void __launch_bounds__(256) k_kv_quantize_v(const bf16_t* __restrict__ k, const bf16_t* __restrict__ vt,
                                                       uint8_t* __restrict__ k8, uint8_t* __restrict__ v8,
                                                       float* __restrict__ ksc, float* __restrict__ vsc, int src_t, int dst_t,
                                                       int t0, int n_tok) {
  constexpr int HD = 96;
  __shared__ unsigned vmax_bits[64];
  const int bh = blockIdx.y, tt = blockIdx.x * 64, tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) vmax_bits[tid] = 0u;
  // ---- V^T loads first (3 chunks of 8 tokens per thread): thread -> (dim d, token chunk c)
  u32x4_t vv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = j * 256 + tid, d = i >> 3, c = i & 7;
    vv[j] = *(const u32x4_t*)(vt + ((size_t)bh * HD + d) * src_t + t0 + tt + 8 * c);
  }
  // ---- K: 4 passes of 16 rows; lane group of 16 = one row, lanes 0..11 hold its 12 chunks
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 16 + (tid >> 4), c = tid & 15, t = t0 + tt + r;
    const bool live = c < 12 && tt + r < n_tok;
    u32x4_t w = {0u, 0u, 0u, 0u};
    if (live) w = *(const u32x4_t*)(k + ((size_t)bh * src_t + t) * HD + c * 8);
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(w[j])), fabsf(bf16hi(w[j]))));
    amax = fmaxf(amax, P3V_DPP_F32(amax, 0xB1));
    amax = fmaxf(amax, P3V_DPP_F32(amax, 0x4E));
    amax = fmaxf(amax, P3V_DPP_F32(amax, 0x124));
    amax = fmaxf(amax, P3V_DPP_F32(amax, 0x128));                // maximum over the 16-lane row
    const float sc = amax > 0.f ? amax / 127.f : 1.f, inv = 1.f / sc;
    if (live) {
      u32x2_t o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q0 = (int)rintf(bf16lo(w[2 * h]) * inv) + 128, q1 = (int)rintf(bf16hi(w[2 * h]) * inv) + 128;
        const int q2 = (int)rintf(bf16lo(w[2 * h + 1]) * inv) + 128, q3 = (int)rintf(bf16hi(w[2 * h + 1]) * inv) + 128;
        o[h] = (uint32_t)q0 | ((uint32_t)q1 << 8) | ((uint32_t)q2 << 16) | ((uint32_t)q3 << 24);
      }
      *(u32x2_t*)(k8 + ((size_t)bh * dst_t + t) * HD + c * 8) = o;
      if (c == 0) ksc[(size_t)bh * dst_t + t] = sc;
    }
  }
  // ---- V^T: per-token maximum over d
  __syncthreads();                                               // vmax_bits cleared
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = (j * 256 + tid) & 7;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicMax(&vmax_bits[8 * c + 2 * e], __float_as_uint(fabsf(bf16lo(vv[j][e]))));
      atomicMax(&vmax_bits[8 * c + 2 * e + 1], __float_as_uint(fabsf(bf16hi(vv[j][e]))));
    }
  }
  __syncthreads();
  if (tid < 64 && tt + tid < n_tok) {
    const float m = __uint_as_float(vmax_bits[tid]);
    vsc[(size_t)bh * dst_t + t0 + tt + tid] = m > 0.f ? m / 127.f : 1.f;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = j * 256 + tid, d = i >> 3, c = i & 7;
    uint32_t o[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float m = __uint_as_float(vmax_bits[8 * c + e]);
      const float sc = m > 0.f ? m / 127.f : 1.f;
      const float x = (e & 1) ? bf16hi(vv[j][e >> 1]) : bf16lo(vv[j][e >> 1]);
      o[e >> 2] |= (uint32_t)(((int)rintf(x / sc) + 128) & 0xff) << (8 * (e & 3));
    }
    uint8_t* dst = v8 + ((size_t)bh * HD + d) * dst_t + t0 + tt + 8 * c;
    if (tt + 8 * c + 8 <= n_tok) {
      *(u32x2_t*)dst = (u32x2_t){o[0], o[1]};
    } else {
      for (int e = 0; e < 8; ++e)
        if (tt + 8 * c + e < n_tok) dst[e] = (uint8_t)(o[e >> 2] >> (8 * (e & 3)));
    }
  }
}

extern "C" int p3v_kv_quantize(const uint16_t* k, const uint16_t* vt, uint8_t* k8, uint8_t* v8t, float* k_scale,
                               float* v_scale, int BH, int hd, int src_t, int dst_t, int t0, int n_tok, void* stream) {
  if (!k || !vt || !k8 || !v8t || !k_scale || !v_scale || BH <= 0 || hd > 96 || hd % 2 || n_tok < 0) return P3V_ERR_ARG;
  if (n_tok == 0) return P3V_OK;
  const bool old_only = p3v_tuning().kvq_old != 0;     // A/B knob
  const bool whole_tiles = (long)p3v_cdiv(n_tok, 64) * 64 + t0 <= src_t;   // the vector kernel reads whole 64-token tiles
  if (!old_only && hd == 96 && t0 % 8 == 0 && src_t % 8 == 0 && dst_t % 8 == 0 && whole_tiles && !(((uintptr_t)k | (uintptr_t)vt) & 15) &&
      !(((uintptr_t)k8 | (uintptr_t)v8t) & 7))
    hipLaunchKernelGGL(k_kv_quantize_v, dim3(p3v_cdiv(n_tok, 64), BH), dim3(256), 0, (hipStream_t)stream, k, vt, k8, v8t,
                       k_scale, v_scale, src_t, dst_t, t0, n_tok);
  else
    hipLaunchKernelGGL(k_kv_quantize, dim3(p3v_cdiv(n_tok, 64), BH), dim3(256), 0, (hipStream_t)stream, k, vt, k8, v8t,
                       k_scale, v_scale, hd, src_t, dst_t, t0, n_tok);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// int8 caches -> bf16 K [BH, dst_t, hd] / V^T [BH, hd, dst_t], tokens [0, n_tok): (code - 128) * scale, rounded once.
// Used by cached calls with more than 16 new tokens on the quantised cache (constrain() with a long constraint text):
// they attend through the prefill kernel on a dequantised copy of the layer (the reference re-dequantises the whole
// prompt every step, phi.py:534-540).
__global__ void __launch_bounds__(256) k_kv_dequantize(const uint8_t* __restrict__ k8, const uint8_t* __restrict__ v8,
                                                       const float* __restrict__ ksc, const float* __restrict__ vsc,
                                                       bf16_t* __restrict__ k, bf16_t* __restrict__ vt, int hd, int src_t,
                                                       int dst_t, int n_tok) {
  const int bh = blockIdx.y;
  const int n = n_tok * hd;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int t = i / hd, d = i - t * hd;                       // K: token-major
    k[((size_t)bh * dst_t + t) * hd + d] = f32_to_bf16(((float)k8[((size_t)bh * src_t + t) * hd + d] - 128.f) * ksc[(size_t)bh * src_t + t]);
    const int d2 = i / n_tok, t2 = i - d2 * n_tok;              // V^T: dim-major
    vt[((size_t)bh * hd + d2) * dst_t + t2] = f32_to_bf16(((float)v8[((size_t)bh * hd + d2) * src_t + t2] - 128.f) * vsc[(size_t)bh * src_t + t2]);
  }
}

extern "C" int p3v_kv_dequantize(const uint8_t* k8, const uint8_t* v8t, const float* k_scale, const float* v_scale, uint16_t* k,
                                 uint16_t* vt, int BH, int hd, int src_t, int dst_t, int n_tok, void* stream) {
  if (!k8 || !v8t || !k_scale || !v_scale || !k || !vt || BH <= 0 || hd <= 0 || n_tok < 0 || n_tok > src_t || n_tok > dst_t) return P3V_ERR_ARG;
  if (n_tok == 0) return P3V_OK;
  hipLaunchKernelGGL(k_kv_dequantize, dim3(min(p3v_cdiv((long)n_tok * hd, 256), 64), BH), dim3(256), 0, (hipStream_t)stream, k8, v8t,
                     k_scale, v_scale, k, vt, hd, src_t, dst_t, n_tok);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

struct AttnDecQ8P {
  const bf16_t* qkv; const float* cos_t; const float* sin_t; uint8_t* k8; uint8_t* v8; float* ksc; float* vsc;
  const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, nh, nkv, past, cache_t, rope_bstride, n_split;
  float scale;
  int grp, grp_magic;          // heads per kv head and ceil(2^16 / grp) (k_attn_decode_q8s)
  int merge;                   // nonzero: in-launch split merge, as AttnDecP
  bf16_t* out;                 // [B, L, nh * 96] (fused merge only)
  // fused o_proj + residual on e4m3 weights (k_attn_decode128_q8<true> only), as AttnDecP's
  const uint8_t* o_w; const float* o_scale; bf16_t* o_x; bf16_t* o_rearm; int o_n;
  int fo_remap;                // as AttnDecP's (fo_map)
};

// 16 offset-binary bytes -> 16 FP16 values 1024 + byte: a byte dropped into the mantissa of 0x6400 (= 1024.0) is exactly
// 1024 + u, so ONE v_perm_b32 converts TWO codes (the first version went through v_cvt_f32_ubyteN + a bf16 pack: three VALU
// instructions per two codes, and that conversion -- not the bytes -- bounded the int8 decode).  Q and P are rounded to
// fp16 (11 significant bits; bf16 has 8) and the products run on the fp16 MFMA at the bf16 rate.  Neither offset is applied
// per element; both fold out of the dot products:
//   sum_d q[d]*(u-128) = sum_d q[d]*(1024+u) - 1152*sum_d q[d]     and     sum_t P[t]*(u-128) = sum_t P[t]*(1024+u) - 1152*sum_t P[t].
#define Q8_OFF 1152.f
__device__ __forceinline__ float f16_round(float x) { return (float)(_Float16)x; }
__device__ __forceinline__ float f16lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
__device__ __forceinline__ float f16hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
__device__ __forceinline__ uint32_t code_pair_f16(int a, int b) { return (0x6400u | (uint32_t)a) | ((0x6400u | (uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t code_f16(int a) { return (uint16_t)(0x6400u | (uint32_t)a); }
__device__ __forceinline__ void u8x16_to_f16(u32x4_t w, u32x4_t& lo, u32x4_t& hi) {
  uint32_t a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = __builtin_amdgcn_perm(0x64646464u, w[j], 0x04010400u);     // bytes {b0, 0x64, b1, 0x64}: codes 4j, 4j+1
    b[j] = __builtin_amdgcn_perm(0x64646464u, w[j], 0x04030402u);     // bytes {b2, 0x64, b3, 0x64}: codes 4j+2, 4j+3
  }
  lo = (u32x4_t){a[0], b[0], a[1], b[1]};
  hi = (u32x4_t){a[2], b[2], a[3], b[3]};
}

__global__ void __launch_bounds__(64) k_attn_decode_q8(AttnDecQ8P p) {
  constexpr int HD = 96, TK = 64, KSTR = HD * 2 + 16, VSTR = TK * 2 + 16, NKS = 3, NDT = 6, NLD = 6;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[TK * KSTR];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[HD * VSTR];
  __shared__ __attribute__((aligned(16))) float ksl[TK], vsl[TK];
  __shared__ __attribute__((aligned(16))) unsigned char Kx[HD * 2], Vx[HD * 2];   // exact new row (quantiser input)
  const int lane = threadIdx.x, g = lane >> 4, qi = lane & 15;
  const int b = blockIdx.z, head = blockIdx.y, kvh = head / (p.nh / p.nkv);
  const bool kv_writer = head % (p.nh / p.nkv) == 0;
  const float sc2 = p.scale * 1.4426950408889634f;
  const int row_w = (p.nh + 2 * p.nkv) * HD;
  const size_t bh = (size_t)b * p.nkv + kvh;
  uint8_t* kc = p.k8 + bh * (size_t)p.cache_t * HD;
  uint8_t* vc = p.v8 + bh * (size_t)HD * p.cache_t;
  float* ksc = p.ksc + bh * p.cache_t;
  float* vsc = p.vsc + bh * p.cache_t;
  const int chunk = ((p.cache_t + p.n_split - 1) / p.n_split + TK - 1) & ~(TK - 1);
  const int kv_lo = blockIdx.x * chunk, kv_hi = min(p.cache_t, kv_lo + chunk);

  u32x4_t kreg[NLD], vreg[NLD];
  float ks_r, vs_r;
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = it * 64 + lane;
      kreg[it] = __builtin_nontemporal_load((const u32x4_t*)(kc + (size_t)(kv0 + i / 6) * HD + (i % 6) * 16));
      vreg[it] = __builtin_nontemporal_load((const u32x4_t*)(vc + (size_t)(i >> 2) * p.cache_t + kv0 + (i & 3) * 16));
    }
    ks_r = ksc[kv0 + lane];
    vs_r = vsc[kv0 + lane];
  };
  if (kv_lo < kv_hi) load_tile(kv_lo);

  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b] : 0;
  int kv_begin = kv_lo;
  const int kv_end = min(total, kv_hi);
  if (pad > kv_begin) kv_begin = pad & ~(TK - 1);
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);

  auto patch_new = [&](int kv0) {                             // exact new rows into LDS (scale 1), quantised copy to the cache
    const int n0 = max(past, kv0), n1 = min(kv_end, kv0 + TK);
#pragma unroll 1
    for (int t = n0; t < n1; ++t) {
      const int r = t - past;
      const bf16_t* row = p.qkv + ((size_t)b * p.L + r) * row_w;
      if (lane < 12) {
        const u32x4_t kn = rope_chunk(row + (p.nh + kvh) * HD, lane, cos_b + r * (HD / 2), sin_b + r * (HD / 2));
        *(u32x4_t*)(Kx + lane * 16) = kn;                     // exact copy for the quantiser below
      }
      for (int d = lane; d < HD; d += 64) *(bf16_t*)(Vx + d * 2) = row[(p.nh + p.nkv + kvh) * HD + d];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // quantise the new row / column; the step itself attends over the SAME quantised values it stores
      // (keeps one code path in the tile: bytes + scale), phi.py:545-546
      const float ka = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + 4 * lane)) : 0.f;
      const float kb = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + 4 * lane + 2)) : 0.f;
      const float kmax = wave_max(fmaxf(fabsf(ka), fabsf(kb)));
      const float s = kmax > 0.f ? kmax / 127.f : 1.f, inv = 1.f / s;
      const int qa = (int)rintf(ka * inv) + 128, qb = (int)rintf(kb * inv) + 128;
      const float va = bf16_to_f32(*(const bf16_t*)(Vx + lane * 2));
      const float vb = lane < 32 ? bf16_to_f32(*(const bf16_t*)(Vx + (lane + 64) * 2)) : 0.f;
      const float vmx = wave_max(fmaxf(fabsf(va), fabsf(vb)));
      const float sv = vmx > 0.f ? vmx / 127.f : 1.f, invv = 1.f / sv;
      const int qva = (int)rintf(va * invv) + 128, qvb = (int)rintf(vb * invv) + 128;
      if (lane < 48) *(uint32_t*)(Ks + (t - kv0) * KSTR + 4 * lane) = code_pair_f16(qa, qb);
      *(uint16_t*)(Vt + lane * VSTR + (t - kv0) * 2) = code_f16(qva);
      if (lane < 32) *(uint16_t*)(Vt + (lane + 64) * VSTR + (t - kv0) * 2) = code_f16(qvb);
      if (lane == 0) { ksl[t - kv0] = s; vsl[t - kv0] = sv; }
      if (kv_writer) {
        if (lane < 48) *(uint16_t*)(kc + (size_t)t * HD + 2 * lane) = (uint16_t)(qa | (qb << 8));
        vc[(size_t)lane * p.cache_t + t] = (uint8_t)qva;
        if (lane < 32) vc[(size_t)(lane + 64) * p.cache_t + t] = (uint8_t)qvb;
        if (lane == 0) { ksc[t] = s; vsc[t] = sv; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  if (kv_begin > kv_lo && kv_begin < kv_end) load_tile(kv_begin);

  const bool qvalid = qi < p.L;
  const int qpos = past + qi;
  f16x8_t qf[NKS];
  {
    const bf16_t* qrow = p.qkv + ((size_t)b * p.L + (qvalid ? qi : 0)) * row_w + head * HD;
    const float* ct = cos_b + (qvalid ? qi : 0) * (HD / 2);
    const float* st = sin_b + (qvalid ? qi : 0) * (HD / 2);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4_t v = rope_chunk<true>(qrow, 4 * ks + g, ct, st);
      if (!qvalid) v = (u32x4_t){0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(f16x8_t, v);
    }
  }
  // 1152 * sum_d q[d] of this lane's query (the folded K offsets): the lane holds 3 x 8 of the 96 values
  float qoff = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const u32x4_t qw = __builtin_bit_cast(u32x4_t, qf[ks]);
#pragma unroll
    for (int j = 0; j < 4; ++j) qoff += f16lo(qw[j]) + f16hi(qw[j]);
  }
  qoff = rows_sum(qoff);
  qoff *= Q8_OFF;
  float m_run = -INFINITY, l_run = 0.f, p_run = 0.f;          // p_run = sum_t P'[t] (the folded V offset)
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int kv0 = kv_begin; kv0 < kv_end; kv0 += TK) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {                        // bytes -> fp16 (1024 + code) LDS images, laid out as the bf16 kernel's
      const int i = it * 64 + lane;
      u32x4_t lo, hi;
      u8x16_to_f16(kreg[it], lo, hi);
      *(u32x4_t*)(Ks + (i / 6) * KSTR + (i % 6) * 32) = lo;
      *(u32x4_t*)(Ks + (i / 6) * KSTR + (i % 6) * 32 + 16) = hi;
      u8x16_to_f16(vreg[it], lo, hi);
      *(u32x4_t*)(Vt + (i >> 2) * VSTR + (i & 3) * 32) = lo;
      *(u32x4_t*)(Vt + (i >> 2) * VSTR + (i & 3) * 32 + 16) = hi;
    }
    ksl[lane] = ks_r;
    vsl[lane] = vs_r;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kv0 + TK > past) {
      patch_new(kv0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (kv0 + TK < kv_end) load_tile(kv0 + TK);

    f32x4_t s[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s[st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const f16x8_t kf = *(const f16x8_t*)(Ks + (16 * st + qi) * KSTR + (32 * ks + 8 * g) * 2);
        s[st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[st], 0, 0, 0);
      }
    }
    float m_t = -INFINITY;
    f32x4_t vsv[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const f32x4_t kk = *(const f32x4_t*)(ksl + 16 * st + 4 * g);
      vsv[st] = *(const f32x4_t*)(vsl + 16 * st + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + 16 * st + 4 * g + r;
        const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
        const float v = vis ? (s[st][r] - qoff) * kk[r] * sc2 : -INFINITY;
        s[st][r] = v;
        m_t = fmaxf(m_t, v);
      }
    }
    m_t = rows_max(m_t);
    const float m_new = fmaxf(m_run, m_t);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float l_t = 0.f, p_t = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[st][r] - m_use);
        l_t += e;
        const float pv = f16_round(e > 0.f ? e * vsv[st][r] : 0.f);    // V scale folded into P (masked keys stay exactly 0)
        s[st][r] = pv;
        p_t += pv;
      }
    l_t = rows_sum(l_t);
    p_t = rows_sum(p_t);
    l_run = l_run * alpha + l_t;
    p_run = p_run * alpha + p_t;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[d] *= alpha;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4_t pw;
      pw[0] = pack_f16x2(s[2 * st][0], s[2 * st][1]);
      pw[1] = pack_f16x2(s[2 * st][2], s[2 * st][3]);
      pw[2] = pack_f16x2(s[2 * st + 1][0], s[2 * st + 1][1]);
      pw[3] = pack_f16x2(s[2 * st + 1][2], s[2 * st + 1][3]);
      const f16x8_t pf = __builtin_bit_cast(f16x8_t, pw);
#pragma unroll
      for (int d = 0; d < NDT; ++d) {
        const unsigned char* vr = Vt + (16 * d + qi) * VSTR + (32 * st + 4 * g) * 2;
        const u32x2_t a0 = *(const u32x2_t*)vr, a1 = *(const u32x2_t*)(vr + 32);
        const u32x4_t aw = {a0[0], a0[1], a1[0], a1[1]};
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, aw), pf, o[d], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (!qvalid) return;
  float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + blockIdx.x) * 16 + qi) * (HD + 2);
  const float voff = Q8_OFF * p_run;
#pragma unroll
  for (int d = 0; d < NDT; ++d) *(f32x4_t*)(w + 16 * d + 4 * g) = o[d] - voff;
  if (g == 0) { w[HD] = m_run; w[HD + 1] = l_run; }
}

// Single-tile int8-KV decode attention: the int8 twin of k_attn_decode (4 waves x one 64-key tile, wave w owns keys
// 16w..16w+15 end to end).  Per wave the K slice is 16 rows x 96 bytes and the V^T slice 96 rows x 16 bytes: 4 loads
// per lane, converted to the RAW byte values as bf16 (exact) and written to the same swizzled LDS images as the bf16
// kernel, so the MFMA part is identical; scales and the -128 offsets are applied to the 4 accumulator values per lane
// (see u8x16_to_f16).  The L new rows are rotated exactly, parked in LDS, quantised one row per wave (the step
// attends over the values it stores, phi.py:545-546) and patched into the tile + cache.
__global__ void __launch_bounds__(256) k_attn_decode_q8s(AttnDecQ8P p) {
  constexpr int TK = 64, WK = 16, HD = 96, KROW = HD * 2, VROW = WK * 2, NKS = 3, NDT = 6, CPR = 12;
  constexpr int KS_BYTES = WK * KROW, VS_BYTES = HD * VROW, WREG = KS_BYTES + VS_BYTES;
  static_assert(WREG >= 16 * HD * 4, "a wave's O partial reuses its tile region");
  __shared__ __attribute__((aligned(16))) unsigned char KV[4 * WREG];   // [wave]{K slice | V^T slice}, bf16 images
  __shared__ __attribute__((aligned(16))) unsigned char Qs[16 * KROW];
  __shared__ __attribute__((aligned(16))) unsigned char Kx[16 * KROW], Vx[16 * KROW];   // exact new rows [r][96] bf16
  __shared__ __attribute__((aligned(16))) float ksl[TK], vsl[TK];
  __shared__ float Ml[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, head = blockIdx.y, kvh = (head * p.grp_magic) >> 16;
  const bool kv_writer = head == kvh * p.grp;
  unsigned char* wreg = KV + wave * WREG;
  const size_t bh = (size_t)b * p.nkv + kvh;
  uint8_t* kc = p.k8 + bh * (size_t)p.cache_t * HD;
  uint8_t* vc = p.v8 + bh * (size_t)HD * p.cache_t;
  float* ksc = p.ksc + bh * p.cache_t;
  float* vsc = p.vsc + bh * p.cache_t;

  int past = p.past, pad = 0;
  if (p.d_past) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(past) : "s"(p.d_past) : "memory");
  if (p.pad_len) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(pad) : "s"(p.pad_len + b) : "memory");

  // ---- this wave's slices of the (static) tile: requested before the cache length has arrived
  const int kv_lo = blockIdx.x * TK, kv_hi = min(p.cache_t, kv_lo + TK);
  const int k0 = min(kv_lo, p.cache_t - TK) + WK * wave;
  const int l32 = lane & 31;
  const u32x4_t k_a = __builtin_nontemporal_load((const u32x4_t*)(kc + (size_t)k0 * HD) + lane);          // chunks 0..63
  const u32x4_t k_b = __builtin_nontemporal_load((const u32x4_t*)(kc + (size_t)k0 * HD) + 64 + l32);      // chunks 64..95
  const u32x4_t v_a = __builtin_nontemporal_load((const u32x4_t*)(vc + (size_t)lane * p.cache_t + k0));        // rows 0..63
  const u32x4_t v_b = __builtin_nontemporal_load((const u32x4_t*)(vc + (size_t)(64 + l32) * p.cache_t + k0)); // rows 64..95
  const float ks_r = ksc[k0 + qi], vs_r = vsc[k0 + qi];

  // ---- the L new rows: Q for this head (rotated -> Qs) and K / V (exact, parked in Kx / Vx for the quantiser)
  const int row_w = (p.nh + 2 * p.nkv) * HD;
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);
  const int tr = tid / CPR, tc = tid - tr * CPR;
  const bool rtask = tr < p.L;
  const int n_vnew = p.L * HD;
  if (tid < 16 * CPR) {
    u32x4_t qv = {0, 0, 0, 0};
    if (rtask) {
      const bf16_t* row = p.qkv + ((size_t)b * p.L + tr) * row_w;
      const RopeRaw qraw = rope_fetch(row + head * HD, tc, cos_b + tr * (HD / 2), sin_b + tr * (HD / 2));
      RopeRaw kraw = qraw;
      const bf16_t* krow = row + (p.nh + kvh) * HD;
      kraw.x0 = *(const u32x4_t*)(krow + tc * 8);
      kraw.x1 = *(const u32x4_t*)(krow + (tc < 6 ? tc * 8 + 48 : tc * 8 - 48));
      qv = rope_apply<true>(qraw, tc);                         // Q: fp16 (MFMA operand); K: the exact bf16 row for the quantiser
      *(u32x4_t*)(Kx + tr * KROW + tc * 16) = rope_apply(kraw, tc);
    }
    *(u32x4_t*)(Qs + tr * KROW + ((tc ^ ((tr >> 2) & 3)) << 4)) = qv;
  }
  {
    const bf16_t* vnew = p.qkv + (size_t)b * p.L * row_w + (p.nh + p.nkv + kvh) * HD;
#pragma unroll 1
    for (int idx = tid; idx < n_vnew; idx += 256) {
      const int r = idx / HD, d = idx - r * HD;
      *(bf16_t*)(Vx + r * KROW + d * 2) = vnew[(size_t)r * row_w + d];
    }
  }

  // ---- dequantise this wave's slices into the bf16 LDS images (raw byte values)
  {
    u32x4_t lo, hi;
    auto put_k = [&](int i, u32x4_t w) {                       // 16-byte chunk i of the slice: row i/6, values 16*(i%6)..+16
      const int row = i / 6, c2 = (i - row * 6) * 2, sw = (row >> 2) & 3;
      u8x16_to_f16(w, lo, hi);
      *(u32x4_t*)(wreg + row * KROW + ((c2 ^ sw) << 4)) = lo;
      *(u32x4_t*)(wreg + row * KROW + (((c2 + 1) ^ sw) << 4)) = hi;
    };
    auto put_v = [&](int d, u32x4_t w) {                       // row d: 16 keys
      const int sw = (d >> 3) & 1;
      u8x16_to_f16(w, lo, hi);
      *(u32x4_t*)(wreg + KS_BYTES + d * VROW + (sw << 4)) = lo;
      *(u32x4_t*)(wreg + KS_BYTES + d * VROW + ((1 ^ sw) << 4)) = hi;
    };
    put_k(lane, k_a);
    put_v(lane, v_a);
    if (lane < 32) { put_k(64 + lane, k_b); put_v(64 + lane, v_b); }
    if (lane < 16) { ksl[WK * wave + lane] = ks_r; vsl[WK * wave + lane] = vs_r; }
  }

  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(past), "+s"(pad)::"memory");
  const int total = past + p.L;
  const int kv_end = min(total, kv_hi);
  const int qpos = past + qi;
  const bool qvalid = qi < p.L;
  const float sc2 = p.scale * 1.4426950408889634f;
  const unsigned k_rd = qi * KROW + ((g ^ ((qi >> 2) & 3)) << 4);
  const unsigned v_rd = KS_BYTES + qi * VROW + (((g >> 1) ^ ((qi >> 3) & 1)) << 4) + (g & 1) * 8;

  float m_run = -INFINITY, l_run = 0.f, p_sum = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int kv0 = kv_lo;
  if (kv0 < kv_end) {
    __syncthreads();                                           // Qs / Kx / Vx complete, every wave's slices are in LDS
    if (kv0 + TK > past) {
      // quantise the new rows that fall in this tile, one row per wave at a time: the tile gets the byte values the
      // cache gets (one code path: bytes + scale)
#pragma unroll 1
      for (int r = wave; r < p.L; r += 4) {
        const int t = past + r, rr = t - kv0;
        if (rr < 0 || rr >= TK || t >= kv_end) continue;
        const float ka = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + r * KROW + 4 * lane)) : 0.f;
        const float kb = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + r * KROW + 4 * lane + 2)) : 0.f;
        const float kmax = wave_max(fmaxf(fabsf(ka), fabsf(kb)));
        const float sk = kmax > 0.f ? kmax / 127.f : 1.f, inv = 1.f / sk;
        const int qa = (int)rintf(ka * inv) + 128, qb = (int)rintf(kb * inv) + 128;
        const float va = bf16_to_f32(*(const bf16_t*)(Vx + r * KROW + lane * 2));
        const float vb = lane < 32 ? bf16_to_f32(*(const bf16_t*)(Vx + r * KROW + (lane + 64) * 2)) : 0.f;
        const float vmx = wave_max(fmaxf(fabsf(va), fabsf(vb)));
        const float sv = vmx > 0.f ? vmx / 127.f : 1.f, invv = 1.f / sv;
        const int qva = (int)rintf(va * invv) + 128, qvb = (int)rintf(vb * invv) + 128;
        unsigned char* dst = KV + (rr >> 4) * WREG;
        const int row = rr & 15, kk = rr & 15;
        if (lane < 48)
          *(uint32_t*)(dst + row * KROW + ((((lane >> 2)) ^ ((row >> 2) & 3)) << 4) + (lane & 3) * 4) = code_pair_f16(qa, qb);
        *(uint16_t*)(dst + KS_BYTES + lane * VROW + (((kk >> 3) ^ ((lane >> 3) & 1)) << 4) + (kk & 7) * 2) = code_f16(qva);
        if (lane < 32)
          *(uint16_t*)(dst + KS_BYTES + (lane + 64) * VROW + (((kk >> 3) ^ (((lane + 64) >> 3) & 1)) << 4) + (kk & 7) * 2) = code_f16(qvb);
        if (lane == 0) { ksl[rr] = sk; vsl[rr] = sv; }
        if (kv_writer) {
          if (lane < 48) *(uint16_t*)(kc + (size_t)t * HD + 2 * lane) = (uint16_t)(qa | (qb << 8));
          vc[(size_t)lane * p.cache_t + t] = (uint8_t)qva;
          if (lane < 32) vc[(size_t)(lane + 64) * p.cache_t + t] = (uint8_t)qvb;
          if (lane == 0) { ksc[t] = sk; vsc[t] = sv; }
        }
      }
      __syncthreads();
    }
    f16x8_t qf[NKS];
    float qoff = 0.f;                                          // 1152 * sum_d q[d] of this lane's query (the folded K offsets)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *(const f16x8_t*)(Qs + k_rd + ks * 64);
      const u32x4_t qw = __builtin_bit_cast(u32x4_t, qf[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j) qoff += f16lo(qw[j]) + f16hi(qw[j]);
    }
    qoff = Q8_OFF * rows_sum(qoff);

    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const f16x8_t kf = *(const f16x8_t*)(wreg + k_rd + ks * 64);
      s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s, 0, 0, 0);
    }
    const f32x4_t kk4 = *(const f32x4_t*)(ksl + WK * wave + 4 * g), vv4 = *(const f32x4_t*)(vsl + WK * wave + 4 * g);
    float m_t = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = kv0 + WK * wave + 4 * g + r;
      const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
      s[r] = vis ? (s[r] - qoff) * kk4[r] * sc2 : -INFINITY;
      m_t = fmaxf(m_t, s[r]);
    }
    m_t = rows_max(m_t);
    const float m_use = m_t == -INFINITY ? 0.f : m_t;
    float l_t = 0.f, p_t = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[r] - m_use);
      l_t += e;
      s[r] = f16_round(e > 0.f ? e * vv4[r] : 0.f);            // V scale folded into P (masked keys stay exactly 0)
      p_t += s[r];
    }
    l_run = rows_sum(l_t);
    p_sum = rows_sum(p_t);
    m_run = m_t;
    const u32x2_t pw = {pack_f16x2(s[0], s[1]), pack_f16x2(s[2], s[3])};
    const f16x4_t pf = __builtin_bit_cast(f16x4_t, pw);
#pragma unroll
    for (int d = 0; d < NDT; ++d) {
      const f16x4_t vf = *(const f16x4_t*)(wreg + v_rd + d * 16 * VROW);
      o[d] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, o[d], 0, 0, 0);
    }
  }

  // ---- merge the four wave partials (as k_attn_decode); the folded V offset leaves here
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                                             // every wave is done reading the tile regions
  if (qvalid) {
    float* Ow = (float*)wreg + qi * HD;
    const float voff = Q8_OFF * p_sum;
#pragma unroll
    for (int d = 0; d < NDT; ++d) *(f32x4_t*)(Ow + 16 * d + 4 * g) = o[d] - voff;
    if (g == 0) { Ml[wave][qi][0] = m_run; Ml[wave][qi][1] = l_run; }
  }
  __syncthreads();
#pragma unroll 1
  for (int idx = tid; idx < n_vnew; idx += 256) {
    const int q = idx / HD, d = idx - q * HD;
    float mk[4], M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mk[k] = Ml[k][q][0]; M = fmaxf(M, mk[k]); }
    const float Mu = M == -INFINITY ? 0.f : M;
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = __builtin_amdgcn_exp2f(mk[k] - Mu);
      acc += c * ((const float*)(KV + k * WREG))[q * HD + d];
      lsum += c * Ml[k][q][1];
    }
    float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + blockIdx.x) * 16 + q) * (HD + 2);
    st_wt(w + d, acc);
    if (d == 0) { st_wt(w + HD, M); st_wt(w + HD + 1, lsum); }
  }
  // fused split merge (as k_attn_decode): the highest split of a (b, head) waits for the others' flags and merges
  if (p.merge && blockIdx.x == p.n_split - 1) {
    __shared__ float merge_scratch[SPLIT_MERGE_SCRATCH(256)];
    split_merge<256>(p.ws + ((size_t)b * p.nh + head) * p.n_split * 16 * (HD + 2), p.out + (size_t)b * p.L * (p.nh * HD) + head * HD,
                     (size_t)p.nh * HD, p.L, p.n_split, nullptr, merge_scratch);
  }
}

// ---- int8 twin of k_attn_decode128 (round 2): 4 waves x one 128-key tile, wave w owns keys [32w, 32w + 32).
// The RAW BYTES go HBM -> LDS by LDS-DMA (K slice: 32 rows x 96 B, contiguous in the cache; V^T tile by rows: 96 x 128 B,
// 16-byte chunk c of row d stored at c ^ (d & 7)) -- half the bytes of the bf16 kernel and no pass through registers --
// and become fp16 operands (1024 + code, see u8x16_to_f16) only when a fragment is read: ds_read_b64 / 2 x ds_read_b32 +
// v_perm_b32.  Scales ride in registers (two float4 per lane and operand).  New rows: rotated exactly, quantised one row
// per wave (the step attends over what it stores, phi.py:545-546), patched into the byte tile + scale registers' LDS
// copy + the cache.  In-launch split merge as the bf16 kernel.
template <bool FO>                                             // FO: + the layer's o_proj (e4m3 weights) + residual, as k_attn_decode128_o
__global__ void __launch_bounds__(256) k_attn_decode128_q8(AttnDecQ8P p) {
  constexpr int TK = 128, WK = 32, HD = 96, KROWB = HD, VROWB = TK, NKS = 3, NDT = 6, CPR = 12, KROW = HD * 2;
  constexpr int KS_BYTES = WK * KROWB;                         // 3 KiB of key bytes per wave
  __shared__ __attribute__((aligned(1024))) unsigned char KV[4 * 6144];   // [K bytes x 4 waves (12 KiB) | V^T bytes (12 KiB)]; later 4 x O partial
  __shared__ __attribute__((aligned(16))) unsigned char Aux[3 * 16 * KROW];   // [Qs | Kx | Vx]; after the tile: [merge scratch | own partial]
  unsigned char* Qs = Aux;                                     // rotated Q, fp16
  unsigned char* Kx = Aux + 16 * KROW;                         // exact new rows [r][96] bf16
  unsigned char* Vx = Aux + 2 * 16 * KROW;
  static_assert(SPLIT_MERGE_SCRATCH(256) * 4 <= 2176 && 2176 + 16 * (HD + 2) * 4 <= 3 * 16 * KROW, "merge scratch + own partial fit the aux block");
  __shared__ __attribute__((aligned(16))) float ksl[TK], vsl[TK];
  __shared__ float Ml[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x, by = blockIdx.y, o_unit = -2, o_unit2 = -1;
  if (FO && p.fo_remap) {                                      // 1-D grid, roles placed by virtual CU (fo_map)
    const FoMap fm = fo_map((int)blockIdx.x, p.n_split, p.nh, p.o_n / 8, p.fo_remap == 2);
    bx = fm.bx, by = fm.by, o_unit = fm.o_unit, o_unit2 = fm.o_unit2;
  }
  const int b = blockIdx.z, head = by, kvh = (head * p.grp_magic) >> 16;
  const bool kv_writer = head == kvh * p.grp;
  unsigned char* kslice = KV + wave * KS_BYTES;
  unsigned char* vtile = KV + 4 * KS_BYTES;
  const size_t bh = (size_t)b * p.nkv + kvh;
  uint8_t* kc = p.k8 + bh * (size_t)p.cache_t * HD;
  uint8_t* vc = p.v8 + bh * (size_t)HD * p.cache_t;
  float* ksc = p.ksc + bh * p.cache_t;
  float* vsc = p.vsc + bh * p.cache_t;

  int past = p.past, pad = 0;
  if (p.d_past) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(past) : "s"(p.d_past) : "memory");
  if (p.pad_len) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(pad) : "s"(p.pad_len + b) : "memory");

  const int kv_lo = bx * TK, kv_hi = min(p.cache_t, kv_lo + TK);
  const int kvd = min(kv_lo, p.cache_t - TK);                  // tile actually fetched (an empty split fetches one it never uses)
  // ---- small loads first: the tile's scales, and the L new rows (Q / K raw for the rotation, V)
  const int row_w = (p.nh + 2 * p.nkv) * HD;
  const int n_vnew = p.L * HD;
  const bf16_t* vnew = p.qkv + (size_t)b * p.L * row_w + (p.nh + p.nkv + kvh) * HD;
  bf16_t v_early = 0;
  if (tid < n_vnew) {
    const unsigned r = (unsigned)tid / HD;
    v_early = vnew[r * (unsigned)row_w + ((unsigned)tid - r * HD)];
  }
  const float scl = tid < TK ? ksc[kvd + tid] : vsc[kvd + tid - TK];
  const float* cos_b = p.cos_t + (size_t)b * p.rope_bstride * (HD / 2);
  const float* sin_b = p.sin_t + (size_t)b * p.rope_bstride * (HD / 2);
  const int tr = tid / CPR, tc = tid - tr * CPR;
  const bool rtask = tr < p.L;
  RopeRaw qraw, kraw;
  if (rtask) {
    const bf16_t* row = p.qkv + ((size_t)b * p.L + tr) * row_w;
    qraw = rope_fetch(row + head * HD, tc, cos_b + tr * (HD / 2), sin_b + tr * (HD / 2));
    kraw = qraw;
    const bf16_t* krow = row + (p.nh + kvh) * HD;
    kraw.x0 = *(const u32x4_t*)(krow + tc * 8);
    kraw.x1 = *(const u32x4_t*)(krow + (tc < 6 ? tc * 8 + 48 : tc * 8 - 48));
  }
  // ---- tile DMA, MUBUF form, AFTER the small loads above (as k_attn_decode128 V2: memory returns in order, and a pending
  // `global_load_lds` makes the compiler wait for everything at every later wait)
  {
    const unsigned char* ksrc = kc + (size_t)(kvd + WK * wave) * KROWB;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)ksrc, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)(vc + kvd), 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < 3; ++j)                                // K bytes: 3 KiB per wave, linear
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (dec_lptr_t)(kslice + j * 1024), 16, (unsigned)(j * 1024 + lane * 16), 0, 0, P3V_ATTN_AUX);
#pragma unroll
    for (int j = 0; j < 3; ++j) {                              // V^T bytes: wave w brings rows 24w..24w+23, 8 rows of 128 B per instruction
      const int d = 24 * wave + 8 * j + (lane >> 3);
      const unsigned voff = (unsigned)d * (unsigned)p.cache_t + ((((unsigned)lane & 7) ^ ((unsigned)d & 7)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (dec_lptr_t)(vtile + wave * 3072 + j * 1024), 16, voff, 0, 0, P3V_ATTN_AUX);
    }
  }
  // ---- their LDS images (the tile is in flight)
  if (tid < TK) ksl[tid] = scl;
  else vsl[tid - TK] = scl;
  if (tid < 16 * CPR) {
    u32x4_t qv = {0, 0, 0, 0};
    if (rtask) {
      qv = rope_apply<true>(qraw, tc);
      *(u32x4_t*)(Kx + tr * KROW + tc * 16) = rope_apply(kraw, tc);
    }
    *(u32x4_t*)(Qs + tr * KROW + ((tc ^ ((tr >> 2) & 3)) << 4)) = qv;
  }
  if (tid < n_vnew) *(bf16_t*)(Vx + (tid / HD) * KROW + (tid % HD) * 2) = v_early;
#pragma unroll 1
  for (int idx = tid + 256; idx < n_vnew; idx += 256) {
    const int r = idx / HD, d = idx - r * HD;
    *(bf16_t*)(Vx + r * KROW + d * 2) = vnew[(size_t)r * row_w + d];
  }

  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(past), "+s"(pad)::"memory");
  const int total = past + p.L;
  const int kv_end = min(total, kv_hi);
  const int qpos = past + qi;
  const bool qvalid = qi < p.L;
  const float sc2 = p.scale * 1.4426950408889634f;
  const unsigned q_rd = qi * KROW + ((g ^ ((qi >> 2) & 3)) << 4);          // + ks*64 (fp16 Q, swizzled as the bf16 kernels')

  float m_run = -INFINITY, l_run = 0.f, p_sum = 0.f;
  f32x4_t o[NDT];
#pragma unroll
  for (int d = 0; d < NDT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int kv0 = kv_lo;
  if (kv0 < kv_end) {
    const bool has_new = kv0 + TK > past;                      // workgroup-uniform: the tile holds new positions
    if (!has_new) {
      // Qs / scales are complete (barrier, no memory wait); this wave's K bytes are its three oldest DMAs: S^T and the
      // softmax run while the V^T bytes are still landing
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's DMA pieces have landed
      __syncthreads();                                         // Qs / Kx / Vx / scales complete, every wave's slices are in LDS
    }
    if (has_new) {
#pragma unroll 1
      for (int r = wave; r < p.L; r += 4) {                    // one new row per wave at a time
        const int t = past + r, rr = t - kv0;
        if (rr < 0 || rr >= TK || t >= kv_end) continue;
        const float ka = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + r * KROW + 4 * lane)) : 0.f;
        const float kb = lane < 48 ? bf16_to_f32(*(const bf16_t*)(Kx + r * KROW + 4 * lane + 2)) : 0.f;
        const float kmax = wave_max(fmaxf(fabsf(ka), fabsf(kb)));
        const float sk = kmax > 0.f ? kmax / 127.f : 1.f, inv = 1.f / sk;
        const int qa = (int)rintf(ka * inv) + 128, qb = (int)rintf(kb * inv) + 128;
        const float va = bf16_to_f32(*(const bf16_t*)(Vx + r * KROW + lane * 2));
        const float vb = lane < 32 ? bf16_to_f32(*(const bf16_t*)(Vx + r * KROW + (lane + 64) * 2)) : 0.f;
        const float vmx = wave_max(fmaxf(fabsf(va), fabsf(vb)));
        const float sv = vmx > 0.f ? vmx / 127.f : 1.f, invv = 1.f / sv;
        const int qva = (int)rintf(va * invv) + 128, qvb = (int)rintf(vb * invv) + 128;
        if (lane < 48) *(uint16_t*)(KV + (rr >> 5) * KS_BYTES + (rr & 31) * KROWB + 2 * lane) = (uint16_t)(qa | (qb << 8));
        vtile[lane * VROWB + ((((rr >> 4)) ^ (lane & 7)) << 4) + (rr & 15)] = (unsigned char)qva;
        if (lane < 32) vtile[(lane + 64) * VROWB + ((((rr >> 4)) ^ ((lane + 64) & 7)) << 4) + (rr & 15)] = (unsigned char)qvb;
        if (lane == 0) { ksl[rr] = sk; vsl[rr] = sv; }
        if (kv_writer) {
          if (lane < 48) *(uint16_t*)(kc + (size_t)t * HD + 2 * lane) = (uint16_t)(qa | (qb << 8));
          vc[(size_t)lane * p.cache_t + t] = (uint8_t)qva;
          if (lane < 32) vc[(size_t)(lane + 64) * p.cache_t + t] = (uint8_t)qvb;
          if (lane == 0) { ksc[t] = sk; vsc[t] = sv; }
        }
      }
      __syncthreads();
    }
    f16x8_t qf[NKS];
    float qoff = 0.f;                                          // 1152 * sum_d q[d] of this lane's query (the folded K offsets)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *(const f16x8_t*)(Qs + q_rd + ks * 64);
      const u32x4_t qw = __builtin_bit_cast(u32x4_t, qf[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j) qoff += f16lo(qw[j]) + f16hi(qw[j]);
    }
    qoff = Q8_OFF * rows_sum(qoff);

    f32x4_t s[2];
    u32x2_t kcode[2][NKS];                                     // key row 16kb + qi, k = 32ks + 8g .. +7: eight code bytes
    if (!has_new) {
      // hidden from the compiler (it would wait for the V^T DMAs too: it cannot tell the two halves of `KV` apart)
      const unsigned ka = (unsigned)(size_t)(dec_lptr_t)kslice + qi * KROWB + 8 * g;
      static_assert(16 * KROWB == 1536, "second key block of the wave's slice");
      asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:32\n\tds_read_b64 %2, %6 offset:64\n\t"
                   "ds_read_b64 %3, %6 offset:1536\n\tds_read_b64 %4, %6 offset:1568\n\tds_read_b64 %5, %6 offset:1600\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(kcode[0][0]), "=&v"(kcode[0][1]), "=&v"(kcode[0][2]), "=&v"(kcode[1][0]), "=&v"(kcode[1][1]), "=&v"(kcode[1][2])
                   : "v"(ka) : "memory");
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) kcode[kb][ks] = *(const u32x2_t*)(kslice + (16 * kb + qi) * KROWB + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const u32x2_t kb8 = kcode[kb][ks];
        const u32x4_t kw = {__builtin_amdgcn_perm(0x64646464u, kb8[0], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, kb8[0], 0x04030402u),
                            __builtin_amdgcn_perm(0x64646464u, kb8[1], 0x04010400u), __builtin_amdgcn_perm(0x64646464u, kb8[1], 0x04030402u)};
        s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, kw), qf[ks], s[kb], 0, 0, 0);
      }
    }
    float m_t = -INFINITY;
    f32x4_t vsv[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const f32x4_t kk = *(const f32x4_t*)(ksl + WK * wave + 16 * kb + 4 * g);
      vsv[kb] = *(const f32x4_t*)(vsl + WK * wave + 16 * kb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + WK * wave + 16 * kb + 4 * g + r;
        const bool vis = t < kv_end && t >= pad && t <= qpos && qpos >= pad;
        s[kb][r] = vis ? (s[kb][r] - qoff) * kk[r] * sc2 : -INFINITY;
        m_t = fmaxf(m_t, s[kb][r]);
      }
    }
    m_t = rows_max(m_t);
    const float m_use = m_t == -INFINITY ? 0.f : m_t;
    float l_t = 0.f, p_t = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kb][r] - m_use);
        l_t += e;
        s[kb][r] = f16_round(e > 0.f ? e * vsv[kb][r] : 0.f);  // V scale folded into P (masked keys stay exactly 0)
        p_t += s[kb][r];
      }
    l_run = rows_sum(l_t);
    p_sum = rows_sum(p_t);
    m_run = m_t;
    const u32x4_t pw = {pack_f16x2(s[0][0], s[0][1]), pack_f16x2(s[0][2], s[0][3]), pack_f16x2(s[1][0], s[1][1]), pack_f16x2(s[1][2], s[1][3])};
    const f16x8_t pf = __builtin_bit_cast(f16x8_t, pw);
    if (!has_new) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // the V^T bytes are every wave's DMA
#pragma unroll
    for (int d = 0; d < NDT; ++d) {                            // V^T row 16d + qi, keys 32w + 16kb + 4g .. +3: four code bytes per block
      const int row = 16 * d + qi;
      const unsigned char* vr = vtile + row * VROWB + 4 * g;
      const uint32_t v0 = *(const uint32_t*)(vr + (((2 * wave) ^ (row & 7)) << 4)), v1 = *(const uint32_t*)(vr + (((2 * wave + 1) ^ (row & 7)) << 4));
      const u32x4_t aw = {__builtin_amdgcn_perm(0x64646464u, v0, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, v0, 0x04030402u),
                          __builtin_amdgcn_perm(0x64646464u, v1, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, v1, 0x04030402u)};
      o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, aw), pf, o[d], 0, 0, 0);
    }
  }

  // ---- merge the four wave partials through the (dead) byte tiles; the folded V offset leaves here
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (qvalid) {
    float* Ow = (float*)(KV + wave * 6144) + qi * HD;
    const float voff = Q8_OFF * p_sum;
#pragma unroll
    for (int d = 0; d < NDT; ++d) *(f32x4_t*)(Ow + 16 * d + 4 * g) = o[d] - voff;
    if (g == 0) { Ml[wave][qi][0] = m_run; Ml[wave][qi][1] = l_run; }
  }
  __syncthreads();
  const bool merger = p.merge && bx == p.n_split - 1;  // the highest split of a (b, head) merges in-launch
  float* own = (float*)(Aux + 2176);                           // its partial: [16][HD + 2] in the (dead) aux block, after the merge scratch
#pragma unroll 1
  for (int idx = tid; idx < n_vnew; idx += 256) {
    const int q = idx / HD, d = idx - q * HD;
    float mk[4], M = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { mk[k] = Ml[k][q][0]; M = fmaxf(M, mk[k]); }
    const float Mu = M == -INFINITY ? 0.f : M;
    float acc = 0.f, lsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float c = __builtin_amdgcn_exp2f(mk[k] - Mu);
      acc += c * ((const float*)(KV + k * 6144))[q * HD + d];
      lsum += c * Ml[k][q][1];
    }
    if (merger) {                                              // stays in LDS: split_merge takes it from there
      own[q * (HD + 2) + d] = acc;
      if (d == 0) { own[q * (HD + 2) + HD] = M; own[q * (HD + 2) + HD + 1] = lsum; }
    } else {
      float* w = p.ws + ((((size_t)b * p.nh + head) * p.n_split + bx) * 16 + q) * (HD + 2);
      st_wt(w + d, acc);
      if (d == 0) { st_wt(w + HD, M); st_wt(w + HD + 1, lsum); }
    }
  }
  if (merger)
    split_merge<256>(p.ws + ((size_t)b * p.nh + head) * p.n_split * 16 * (HD + 2), p.out + (size_t)b * p.L * (p.nh * HD) + head * HD,
                     (size_t)p.nh * HD, p.L, p.n_split, own, (float*)Aux, FO);
  if (FO) {
    if (bx == 0 && by == 0) {                  // re-arm the other layer parity's buffer (nobody reads it in this launch)
      uint32_t* ra = (uint32_t*)p.o_rearm;
      for (int i = tid; i < p.nh * HD / 2; i += 256) ra[i] = 0xffffffffu;
    }
    const int o_wl = o_unit != -2 ? o_unit : by * (p.n_split - 1) + bx;   // projection unit (fo_map), or the rank among the non-merging workgroups
    if (!merger && bx < p.n_split - 1 && o_wl >= 0 && o_wl * 8 < p.o_n)
      fo_project<FO_F8, 2>(FoP{p.o_w, p.o_scale, p.o_x, p.out, p.nh}, o_wl, KV, o_unit2);
  }
}

extern "C" int p3v_attention_decode_q8_can_fuse_oproj(int B, int L, int n_heads, int hd, int n_split, int cache_t, int o_n, int merge_in_launch) {
  if (!merge_in_launch) return combine_o_ok(B, L, n_heads, hd, n_split, o_n) ? 2 : 0;      // the merge launch carries the e4m3 o_proj (k_attn_combine_o)
  if (B != 1 || L != 1 || hd != 96 || n_heads * hd != 3072 || o_n <= 0 || o_n % 8) return 0;
  if (!(n_split * 128 >= cache_t && cache_t % 128 == 0 && n_split * 64 < cache_t) || n_split < 2) return 0;
  if ((long)(n_split - 1) * n_heads * 8 < o_n || p3v_tuning().q8_old) return 0;
  static long capacity_of[P3V_MAX_DEVICES] = {0};              // every workgroup resident at once: see p3v_attention_decode_can_fuse_oproj
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= P3V_MAX_DEVICES) return 0;
  if (!capacity_of[dev]) {
    int per_cu = 0;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_attn_decode128_q8<true>, 256, 0) != hipSuccess || per_cu < 1) return 0;
    capacity_of[dev] = (long)per_cu * (pr.multiProcessorCount - 8);
  }
  const long capacity = capacity_of[dev];
  return (long)B * n_heads * n_split <= capacity;
}

extern "C" int p3v_attention_decode_q8(const p3v_attn_decode_q8_args_t* a, void* stream) {
  if (!a || !a->qkv || !a->cos_t || !a->sin_t || !a->k8 || !a->v8t || !a->k_scale || !a->v_scale || !a->out || !a->ws)
    return P3V_ERR_ARG;
  if (a->hd != 96) return P3V_ERR_UNSUPPORTED;
  if (a->B <= 0 || a->L <= 0 || a->L > P3V_DECODE_MAX_L || a->n_heads % a->n_kv) return P3V_ERR_ARG;
  if (a->n_split < 1 || a->n_split > 128 || a->cache_t % 64) return P3V_ERR_ARG;
  if (((uintptr_t)a->ws | (uintptr_t)a->out) & 15) return P3V_ERR_ARG;      // (16-byte write-through stores, as p3v_attention_decode)
  const int grp = a->n_heads / a->n_kv;
  AttnDecQ8P p = {a->qkv, a->cos_t, a->sin_t, a->k8, a->v8t, a->k_scale, a->v_scale, a->pad_len, a->d_past, a->ws,
                  a->B, a->L, a->n_heads, a->n_kv, a->past, a->cache_t, a->rope_bstride, a->n_split, a->scale,
                  grp, (65536 + grp - 1) / grp, 0, (bf16_t*)a->out, nullptr, nullptr, nullptr, nullptr, 0};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a->n_split, a->n_heads, a->B);
  const bool old_only = p3v_tuning().q8_old != 0;   // A/B knob
  // the o_proj rides either in the one-tile attention launch (in-launch merge) or in the merge launch (k_attn_combine_o, round 6)
  const bool merge_o = a->o_proj_w8 && !a->merge_in_launch;
  if (merge_o) {
    if (!combine_o_ok(a->B, a->L, a->n_heads, a->hd, a->n_split, a->o_n)) return P3V_ERR_UNSUPPORTED;
    if (!a->o_proj_scale || !a->o_proj_x || !a->o_rearm || ((uintptr_t)a->o_proj_w8 | (uintptr_t)a->o_proj_x | (uintptr_t)a->o_rearm) & 15)
      return P3V_ERR_ARG;
  }
  auto merge = [&]() -> int {                                    // the launch that merges the partials
    if (merge_o)
      return launch_combine_o<FO_F8>(a->ws, a->out, a->o_rearm, a->o_proj_w8, a->o_proj_scale, a->o_proj_x, a->n_heads, a->n_split, a->o_n, s);
    hipLaunchKernelGGL(k_attn_combine2, dim3(a->B * a->n_heads * a->L), dim3(combine_threads(a->n_split)), 0, s, a->ws, a->out, a->L, a->n_heads,
                       a->hd, a->n_split);
    P3V_CHECK_LAUNCH();
    return P3V_OK;
  };
  if (a->o_proj_w8 && a->merge_in_launch &&
      !(!old_only && a->cache_t % 128 == 0 && a->n_split * 128 >= a->cache_t && a->n_split * 64 < a->cache_t))
    return P3V_ERR_UNSUPPORTED;                                 // (the in-launch form exists for the 128-key plan only)
  // one tile per workgroup and at most 16 of them (contexts up to 1k): the 4-wave kernel, which with `merge_in_launch` also
  // merges the splits inside the launch.  Beyond that the single-wave kernel + merge launch is as fast or faster
  // (measured at 42 tiles, config 5 decode: 1.543 ms/step against 1.560 with the 4-wave kernel + fused merge: its
  // bytes go through registers and a ds_write pass into the fp16 images, where the bf16 kernel uses LDS-DMA).
  if (!old_only && a->cache_t % 128 == 0 && a->n_split * 128 >= a->cache_t && a->n_split * 64 < a->cache_t) {   // 128-key tiles
    p.merge = a->merge_in_launch;
    if (a->o_proj_w8 && a->merge_in_launch) {                  // attention + o_proj (e4m3) + residual in one launch
      if (!p3v_attention_decode_q8_can_fuse_oproj(a->B, a->L, a->n_heads, a->hd, a->n_split, a->cache_t, a->o_n, a->merge_in_launch))
        return P3V_ERR_UNSUPPORTED;
      if (!a->o_proj_scale || !a->o_proj_x || !a->o_rearm || ((uintptr_t)a->o_proj_w8 | (uintptr_t)a->o_proj_x | (uintptr_t)a->o_rearm) & 15)
        return P3V_ERR_ARG;
      p.o_w = a->o_proj_w8; p.o_scale = a->o_proj_scale; p.o_x = (bf16_t*)a->o_proj_x; p.o_rearm = (bf16_t*)a->o_rearm; p.o_n = a->o_n;
      // (the virtual-CU placement of k_attn_decode128_o, off by default here: with 24 KB tiles and 9.4 MB of e4m3 W_o the (split, head)
      //  grid measured 0.3 - 0.8 % FASTER on config 5 -- 1.197 vs 1.203 ms per step, profiles/r06_attn_oproj_placement.txt)
      if (p3v_tuning().attn_fo_map_q8 && fo_map_ok(a->n_split, a->n_heads, a->o_n)) {
        p.fo_remap = p3v_tuning().attn_fo_map_q8;
        hipLaunchKernelGGL(k_attn_decode128_q8<true>, dim3(a->n_split * a->n_heads, 1, 1), dim3(256), 0, s, p);
        P3V_CHECK_LAUNCH();
        return P3V_OK;
      }
      hipLaunchKernelGGL(k_attn_decode128_q8<true>, grid, dim3(256), 0, s, p);
      P3V_CHECK_LAUNCH();
      return P3V_OK;
    }
    hipLaunchKernelGGL(k_attn_decode128_q8<false>, grid, dim3(256), 0, s, p);
    P3V_CHECK_LAUNCH();
    if (a->merge_in_launch) return P3V_OK;
    return merge();
  }
  const bool single_tile = !old_only && a->n_split * 64 >= a->cache_t && a->n_split <= 16;
  if (single_tile && a->merge_in_launch) {
    p.merge = a->merge_in_launch;
    hipLaunchKernelGGL(k_attn_decode_q8s, grid, dim3(256), 0, s, p);
    P3V_CHECK_LAUNCH();
    return P3V_OK;
  }
  if (single_tile) hipLaunchKernelGGL(k_attn_decode_q8s, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_attn_decode_q8, grid, dim3(64), 0, s, p);
  P3V_CHECK_LAUNCH();
  return merge();
}
