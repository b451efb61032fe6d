// fp8 (OCP e4m3fn) weight-only projections for the decode path (quantize_model=True; BASELINE config 5;
// replaces the reference's int4 group-64 `nn.quantize`, phi_3_vision_mlx.py:264,296 -- SURVEY.md K23).
// Weights: u8 [N, K] e4m3 bit patterns, one fp32 scale per output row (w = fp8 * scale[n]); activations stay
// bf16 and all accumulation is fp32, so the only new error is the weight rounding.  A decode step then
// streams half the bytes: the kernels below are the fp8 twins of k_gemv3 / k_gemv_mfma in p3v_gemv.hip.
// Prefill uses k_dequant_fp8 (fp8 -> bf16 scratch) + the bf16 MFMA GEMM.
#include <stdlib.h>

#include <type_traits>

#include "p3v_common.h"
#include "p3v_dot_f8.h"
#include "p3v_gemv3_body.h"      // GemvStepP + the step-end helpers shared with the bf16 kernel (IC0 / IC1 come from there too)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

struct GemvF8P {
  const bf16_t* x; const uint8_t* W; const float* wscale; void* out; const bf16_t* resid; const bf16_t* norm_w;
  float eps;
  int M, N, K, epi, units;
};

// ---------------------------------------------------------------- M = 1 streaming (see k_gemv3)
// STEP (p3v_gemv3_body.h): STEP_BEGIN / STEP_END carry the replayed greedy step's embedding gather + rotation rows / arg-max + bookkeeping,
// exactly as gemv3_body does for bf16 weights (round 6: config 5's step is 129 launches too)
template <int NST, int CH, int STEP>
__device__ __forceinline__ void gemv3_f8_body(const GemvF8P& p, int units_per_wave, int wpw, unsigned char* smem, float* red, const GemvStepP* sp) {
  constexpr int CHUNKS = NST * CH * 64;                 // 16-byte weight chunks per row (K = 16 * CHUNKS)
  constexpr int XCH = CHUNKS * 2;                       // 16-byte x chunks
  constexpr int XC = (XCH + 255) / 256;
  u32x4_t* xs = (u32x4_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool silu = p.epi == P3V_EPI_SILU_MUL, has_res = p.epi == P3V_EPI_RESID_BF16;
  const int u_begin = wave < wpw ? min(p.units, (blockIdx.x * wpw + wave) * units_per_wave) : p.units;   // (wpw: see p3v_gemv_wpw)
  const int u_end = min(p.units, u_begin + units_per_wave);
  const int n_st = (u_end - u_begin) * NST;

  u32x4_t xv[XC], gv[XC];
  const bf16_t* xrow[1] = {p.x};
  if (STEP == STEP_BEGIN) {
    int id = sp->tok[0];                                // (uniform: a scalar load)
    id = id < 0 ? 0 : (id >= sp->vocab ? sp->vocab - 1 : id);
    xrow[0] = sp->table + (size_t)id * (XCH * 8);
  }
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = min(tid + k * 256, XCH - 1);
    xv[k] = ((const u32x4_t*)xrow[0])[c];
    gv[k] = p.norm_w ? ((const u32x4_t*)p.norm_w)[c] : (u32x4_t){0, 0, 0, 0};
  }
  u32x4_t wbuf[2][2][CH];
  uint32_t rbuf[2];
  float sbuf[2][2];
  auto issue = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int u = min(u_begin + gs / NST, p.units - 1), s = gs % NST;
    const int r0 = silu ? u : 2 * u, r1 = silu ? u + p.N : min(2 * u + 1, p.N - 1);
    const u32x4_t* w0 = (const u32x4_t*)(p.W + (size_t)r0 * (CHUNKS * 16)) + s * CH * 64 + lane;
    const u32x4_t* w1 = (const u32x4_t*)(p.W + (size_t)r1 * (CHUNKS * 16)) + s * CH * 64 + lane;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      wbuf[buf][0][j] = __builtin_nontemporal_load(w0 + j * 64);
      wbuf[buf][1][j] = __builtin_nontemporal_load(w1 + j * 64);
    }
    rbuf[buf] = has_res ? *(const uint32_t*)(p.resid + 2 * u) : 0u;
    sbuf[buf][0] = p.wscale[r0];
    sbuf[buf][1] = p.wscale[r1];
  };
  if (n_st > 0) issue(0, IC0{});

  float r = 1.f;
  if (p.norm_w) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < XC; ++k)
      if (tid + k * 256 < XCH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float a = bf16lo(xv[k][j]), b = bf16hi(xv[k][j]); ss += a * a + b * b; }
      }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    r = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)(CHUNKS * 16) + p.eps);
  }
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = tid + k * 256;
    if (c < XCH) {
      u32x4_t o = xv[k];
      if (p.norm_w) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = rms_pair(xv[k][j], r, gv[k][j]);
      }
      xs[c] = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float a0 = 0.f, a1 = 0.f;
  ArgMaxVI best[1] = {ArgMaxVI{-INFINITY, 0x7fffffff}};      // STEP_END: this wave's arg-max candidate (lane 0's copy counts)
  auto compute = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int s = gs % NST;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = (s * CH + j) * 64 + lane;
      const u32x4_t xa = xs[2 * c], xb = xs[2 * c + 1];
      a0 = dot16_f8(wbuf[buf][0][j], xa, xb, a0);
      a1 = dot16_f8(wbuf[buf][1][j], xa, xb, a1);
    }
    if (s == NST - 1) {
      const int u = u_begin + gs / NST;
      a0 = wave_sum(a0) * sbuf[buf][0];
      a1 = wave_sum(a1) * sbuf[buf][1];
      if (lane == 0) {
        if (silu) {
          const float g = bf16_round(a0), up = bf16_round(a1);
          ((bf16_t*)p.out)[u] = f32_to_bf16(bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * up);
        } else if (p.epi == P3V_EPI_F32) {
          ((float*)p.out)[2 * u] = a0;
          ((float*)p.out)[2 * u + 1] = a1;
        } else {
          float v0 = a0, v1 = a1;
          if (has_res) { v0 = bf16lo(rbuf[buf]) + bf16_round(v0); v1 = bf16hi(rbuf[buf]) + bf16_round(v1); }
          *(uint32_t*)((bf16_t*)p.out + 2 * u) = pack_bf16x2(v0, v1);
          if (STEP == STEP_END) {                              // on the values just stored (bf16)
            amax_take(best[0], bf16_round(v0), 2 * u);
            if (2 * u + 1 < p.N) amax_take(best[0], bf16_round(v1), 2 * u + 1);
          }
        }
      }
      a0 = a1 = 0.f;
    }
  };
  int gs = 0;
  while (gs + 2 < n_st) {
    issue(gs + 1, IC1{}); compute(gs, IC0{});
    issue(gs + 2, IC0{}); compute(gs + 1, IC1{});
    gs += 2;
  }
  if (gs + 1 < n_st) { issue(gs + 1, IC1{}); compute(gs, IC0{}); compute(gs + 1, IC1{}); }
  else if (gs < n_st) compute(gs, IC0{});
  if (STEP == STEP_BEGIN && blockIdx.x == 0) gemv_step_begin_tail<1>(sp, xrow, 1, XCH, tid);
  if (STEP == STEP_END) gemv_step_end_tail<1>(sp, best, 1, (int)blockIdx.x, tid);
}

template <int NST, int CH>
__global__ void __launch_bounds__(256) k_gemv3_f8(GemvF8P p, int units_per_wave, int wpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4];
  gemv3_f8_body<NST, CH, STEP_NONE>(p, units_per_wave, wpw, smem, red, nullptr);
}

template <int NST, int CH, int STEP>
__global__ void __launch_bounds__(256) k_gemv3_f8_step(GemvF8P p, int units_per_wave, int wpw, GemvStepP sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4];
  gemv3_f8_body<NST, CH, STEP>(p, units_per_wave, wpw, smem, red, &sp);
}

// ---------------------------------------------------------------- 2 <= M <= 16 through the matrix cores (see k_gemv_mfma)
// a 16-byte weight load = 16 consecutive k of one row = the A fragments of TWO MFMAs; the MFMA k index is
// permuted accordingly (lane group g owns k = 64*ds + 16*g .. +15), identically on the x side.
#define F8_G 4                          // double-steps (64 k) per pipeline stage

__device__ __forceinline__ void f8x16_to_bf16(u32x4_t w, u32x4_t& lo, u32x4_t& hi) {
  float f[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], true);
    f[4 * q] = a[0]; f[4 * q + 1] = a[1]; f[4 * q + 2] = b[0]; f[4 * q + 3] = b[1];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { lo[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]); hi[j] = pack_bf16x2(f[8 + 2 * j], f[9 + 2 * j]); }
}

template <bool SILU>
__global__ void __launch_bounds__(256) k_gemv_mfma_f8(GemvF8P p) {
  __shared__ float cpart[4][2][256];
  __shared__ float sspart[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int n_base = blockIdx.x * 16;
  const int kq = p.K >> 2, k_lo = wave * kq, n_st = kq / (64 * F8_G);
  const int n_row = min(n_base + li, p.N - 1);
  const uint8_t* w0 = p.W + (size_t)n_row * p.K + k_lo + 16 * g;
  const uint8_t* w1 = p.W + (size_t)(n_row + p.N) * p.K + k_lo + 16 * g;
  const bf16_t* xp = p.x + (size_t)min(li, p.M - 1) * p.K + k_lo + 16 * g;
  const bf16_t* gp = p.norm_w ? p.norm_w + k_lo + 16 * g : nullptr;

  u32x4_t wa[2][F8_G], wb[2][F8_G], xa[2][2 * F8_G], ga[2][2 * F8_G];
  auto issue = [&](int st, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int ds = 0; ds < F8_G; ++ds) {
      const int off = (st * F8_G + ds) * 64;
      wa[buf][ds] = __builtin_nontemporal_load((const u32x4_t*)(w0 + off));
      if (SILU) wb[buf][ds] = __builtin_nontemporal_load((const u32x4_t*)(w1 + off));
      xa[buf][2 * ds] = *(const u32x4_t*)(xp + off);
      xa[buf][2 * ds + 1] = *(const u32x4_t*)(xp + off + 8);
      if (gp) { ga[buf][2 * ds] = *(const u32x4_t*)(gp + off); ga[buf][2 * ds + 1] = *(const u32x4_t*)(gp + off + 8); }
    }
  };
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;
  auto compute = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int ds = 0; ds < F8_G; ++ds) {
      u32x4_t a_lo, a_hi, b_lo, b_hi;
      f8x16_to_bf16(wa[buf][ds], a_lo, a_hi);
      if (SILU) f8x16_to_bf16(wb[buf][ds], b_lo, b_hi);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4_t xv = xa[buf][2 * ds + h];
        if (gp) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = bf16lo(xv[j]), b = bf16hi(xv[j]);
            ss += a * a + b * b;
            xv[j] = pack_bf16x2(a * bf16lo(ga[buf][2 * ds + h][j]), b * bf16hi(ga[buf][2 * ds + h][j]));
          }
        }
        const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xv);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, h ? a_hi : a_lo), xb, acc0, 0, 0, 0);
        if (SILU) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, h ? b_hi : b_lo), xb, acc1, 0, 0, 0);
      }
    }
  };
  issue(0, IC0{});
  int st = 0;
  while (st + 2 < n_st) {
    issue(st + 1, IC1{}); compute(IC0{});
    issue(st + 2, IC0{}); compute(IC1{});
    st += 2;
  }
  if (st + 1 < n_st) { issue(st + 1, IC1{}); compute(IC0{}); compute(IC1{}); }
  else compute(IC0{});

  ss = rows_sum(ss);
  if (g == 0) sspart[wave][li] = ss;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    cpart[wave][0][lane * 4 + r] = acc0[r];
    if (SILU) cpart[wave][1][lane * 4 + r] = acc1[r];
  }
  __syncthreads();
  if (wave != 0 || li >= p.M) return;
  const float rs = gp ? rsqrtf(((sspart[0][li] + sspart[1][li]) + (sspart[2][li] + sspart[3][li])) / (float)p.K + p.eps) : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n_base + 4 * g + r, e = lane * 4 + r;
    if (n >= p.N) continue;
    const float v0 = ((cpart[0][0][e] + cpart[1][0][e]) + (cpart[2][0][e] + cpart[3][0][e])) * rs * p.wscale[n];
    const size_t o = (size_t)li * p.N + n;
    if (SILU) {
      const float v1 = ((cpart[0][1][e] + cpart[1][1][e]) + (cpart[2][1][e] + cpart[3][1][e])) * rs * p.wscale[n + p.N];
      const float gt = bf16_round(v0), up = bf16_round(v1);
      ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
    } else if (p.epi == P3V_EPI_F32) {
      ((float*)p.out)[o] = v0;
    } else if (p.epi == P3V_EPI_RESID_BF16) {
      ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32(p.resid[o]) + bf16_round(v0));
    } else {
      ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
    }
  }
}

// ---------------------------------------------------------------- 5 <= M <= 8 rows: e4m3 twin of k_gemv_mfma8 (p3v_gemv.hip)
// Same three ideas: whole 128-byte lines per load instruction (MFMA row i = 2r + h: weight row r of an 8-row set, lane (i, g)
// loads the 16 bytes = 16 weights of chunk 2g + h; one load feeds the two MFMAs of its k-halves), activations staged once per
// workgroup in LDS as bf16(x * r * g), and the wave's whole weight slice (12 loads at K = 3072) requested up front.  The
// k_gemv_mfma_f8 stream above issues FIVE 16-byte loads per lane (weights, 2 x, 2 norm weights) for each one from HBM.
template <bool SILU, int NW, int NL>                     // NL = 128-element lines per wave slice
__global__ void __launch_bounds__(NW * 64) k_gemv_mfma8_f8(GemvF8P p) {
  constexpr int KQ = NL * 128, K = KQ * NW, XS = K * 2 + 64, NCH = KQ / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[NW][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int r8 = (lane & 15) >> 1, chunk = 2 * g + (lane & 1);
  const int n_base = blockIdx.x * (SILU ? 8 : 16), k_lo = wave * KQ;
  const int row0 = min(n_base + r8, p.N - 1);
  const int row1 = SILU ? p.N + row0 : min(n_base + 8 + r8, p.N - 1);
  const uint8_t* w0 = p.W + (size_t)row0 * K + k_lo + 16 * chunk;
  const uint8_t* w1 = p.W + (size_t)row1 * K + k_lo + 16 * chunk;

  unsigned char* xslice = smem + k_lo * 2;
  constexpr int XP = (NCH + 63) / 64;                    // x chunks per lane and row (2 at K = 3072, 8192)
  u32x4_t gv[XP], wa[NL], wb[NL];
  float ss[8];
  {
    u32x4_t xv[8][XP];
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int c = min(lane + 64 * k, NCH - 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) xv[m][k] = *(const u32x4_t*)(p.x + (size_t)min(m, p.M - 1) * K + k_lo + 8 * c);
      gv[k] = p.norm_w ? *(const u32x4_t*)(p.norm_w + k_lo + 8 * c) : (u32x4_t){0, 0, 0, 0};
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      wa[l] = __builtin_nontemporal_load((const u32x4_t*)(w0 + l * 128));
      wb[l] = __builtin_nontemporal_load((const u32x4_t*)(w1 + l * 128));
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      ss[m] = 0.f;
#pragma unroll
      for (int k = 0; k < XP; ++k) {
        const int c = lane + 64 * k;
        if (c < NCH) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float a = bf16lo(xv[m][k][j]), b = bf16hi(xv[m][k][j]); ss[m] += a * a + b * b; }
          *(u32x4_t*)(xslice + m * XS + c * 16) = m < p.M ? xv[m][k] : (u32x4_t){0, 0, 0, 0};
        }
      }
    }
  }
  if (p.norm_w) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float t = wave_sum(ss[m]);
      if (lane == 0) red[wave][m] = t;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w][m];
      const float r = rsqrtf(t / (float)K + p.eps);
#pragma unroll
      for (int k = 0; k < XP; ++k) {
        const int c = lane + 64 * k;
        if (c < NCH) {
          u32x4_t* px = (u32x4_t*)(xslice + m * XS + c * 16);
          const u32x4_t v = *px;
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = rms_pair(v[j], r, gv[k][j]);
          *px = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // the slice is wave-private: no barrier

  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* xrow = xslice + r8 * XS + chunk * 32;                     // 16 weights <-> 32 bytes of bf16 x
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    u32x4_t a_lo, a_hi, b_lo, b_hi;
    f8x16_to_bf16(wa[l], a_lo, a_hi);
    f8x16_to_bf16(wb[l], b_lo, b_hi);
    const bf16x8_t x0 = *(const bf16x8_t*)(xrow + l * 256), x1 = *(const bf16x8_t*)(xrow + l * 256 + 16);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_lo), x0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b_lo), x0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_hi), x1, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b_hi), x1, acc1, 0, 0, 0);
  }

  const bool odd = lane & 1;
  float e[2][2] = {{odd ? acc0[1] : acc0[0], odd ? acc0[3] : acc0[2]}, {odd ? acc1[1] : acc1[0], odd ? acc1[3] : acc1[2]}};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) e[t][u] += P3V_DPP_F32(e[t][u], 0xB1);          // quad_perm [1,0,3,2]: the other k-half
  __syncthreads();
  float* cpart = (float*)smem;                                        // [NW][2 sets][8 rows][8 x rows]
  if (!odd) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) cpart[((wave * 2 + t) * 8 + 2 * g + u) * 8 + r8] = e[t][u];
  }
  __syncthreads();
  if (tid >= (SILU ? 64 : 128)) return;
  const int R = tid & 7, m = (tid >> 3) & 7, set = tid >> 6;
  const int n = n_base + set * 8 + R;
  if (m >= p.M || n >= p.N) return;
  float v0 = 0.f, v1 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    v0 += cpart[((w * 2 + set) * 8 + R) * 8 + m];
    if (SILU) v1 += cpart[((w * 2 + 1) * 8 + R) * 8 + m];
  }
  v0 *= p.wscale[n];
  const size_t o = (size_t)m * p.N + n;
  if (SILU) {
    v1 *= p.wscale[n + p.N];
    const float gt = bf16_round(v0), up = bf16_round(v1);
    ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
  } else if (p.epi == P3V_EPI_F32) {
    ((float*)p.out)[o] = v0;
  } else if (p.epi == P3V_EPI_RESID_BF16) {
    ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32(p.resid[o]) + bf16_round(v0));
  } else {
    ((bf16_t*)p.out)[o] = f32_to_bf16(v0);
  }
}

template <bool SILU, int NW, int NL>
static int launch_gemv_mfma8_f8(const GemvF8P& p, hipStream_t s) {
  const size_t lds = (size_t)8 * (p.K * 2 + 64);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k_gemv_mfma8_f8<SILU, NW, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemv_mfma8_f8<SILU, NW, NL>), dim3(p3v_cdiv(p.N, SILU ? 8 : 16)), dim3(NW * 64), lds, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// ---------------------------------------------------------------- fp8 -> bf16 (prefill scratch)
__global__ void __launch_bounds__(256) k_dequant_fp8(const u32x4_t* __restrict__ w8, const float* __restrict__ scale,
                                                     u32x4_t* __restrict__ out, int chunks_per_row, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float s = scale[i / chunks_per_row];
  const u32x4_t w = w8[i];
  float f[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], true);
    f[4 * q] = a[0] * s; f[4 * q + 1] = a[1] * s; f[4 * q + 2] = b[0] * s; f[4 * q + 3] = b[1] * s;
  }
  u32x4_t lo, hi;
#pragma unroll
  for (int j = 0; j < 4; ++j) { lo[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]); hi[j] = pack_bf16x2(f[8 + 2 * j], f[9 + 2 * j]); }
  out[2 * i] = lo;
  out[2 * i + 1] = hi;
}

extern "C" int p3v_dequant_fp8(const uint8_t* w8, const float* scale, uint16_t* out_bf16, int rows, int K, void* stream) {
  if (!w8 || !scale || !out_bf16 || rows <= 0 || K <= 0 || K % 16) return P3V_ERR_ARG;
  const long total = (long)rows * (K / 16);
  hipLaunchKernelGGL(k_dequant_fp8, dim3(p3v_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)w8, scale,
                     (u32x4_t*)out_bf16, K / 16, total);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <int NST, int CH, int STEP = STEP_NONE>
static int launch_gemv3_f8(const GemvF8P& p, hipStream_t s, const GemvStepP* sp = nullptr) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return P3V_ERR_HIP;
    n_cu = pr.multiProcessorCount;
  }
  const int wpc = p3v_tuning().gemv_f8_wpc;   // waves per CU (a stage is half the bytes of the bf16 kernel's: twice the waves keep as many in flight; 8: 1.321, 16: 1.301, 24: 1.333 ms/step)
  int upw = p3v_cdiv(p.units, n_cu * wpc);
  if (upw < 1) upw = 1;
  const int waves = p3v_cdiv(p.units, upw);
  const int wpw = p3v_gemv_wpw(waves, n_cu, p3v_tuning().gemv_wpw);
  if constexpr (STEP != STEP_NONE) {
    if (p3v_cdiv(waves, wpw) > P3V_GEMV_STEP_MAX_WG) return P3V_ERR_UNSUPPORTED;        // (amax_ws holds one candidate per workgroup)
    hipLaunchKernelGGL((k_gemv3_f8_step<NST, CH, STEP>), dim3(p3v_cdiv(waves, wpw)), dim3(256), (size_t)p.K * 2, s, p, upw, wpw, *sp);
  } else {
    hipLaunchKernelGGL((k_gemv3_f8<NST, CH>), dim3(p3v_cdiv(waves, wpw)), dim3(256), (size_t)p.K * 2, s, p, upw, wpw);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// p3v_gemv_step on e4m3 weights: the first / last projection of a replayed greedy step with p3v_step_begin / p3v_step_end folded in
// (one row, K = 3072 or 8192, no epilogue); anything else reports P3V_ERR_UNSUPPORTED and the caller keeps the separate launches.
extern "C" int p3v_gemv_fp8_step(const p3v_gemv_fp8_args_t* a, const p3v_gemv_step_t* st, void* stream) {
  if (!a || !st || !a->W || !a->w_scale || !a->out) return P3V_ERR_ARG;
  const bool begin = st->tok != nullptr, end = st->next_tok != nullptr;
  if (begin == end) return P3V_ERR_ARG;                        // exactly one of the two ends
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return P3V_ERR_ARG;
  if (a->M != 1 || a->N % 2 || (a->K != 3072 && a->K != 8192) || a->epilogue != P3V_EPI_NONE) return P3V_ERR_UNSUPPORTED;
  if (begin) {
    if (!st->embed_table || !st->x_out || !st->cos_t || !st->sin_t || !st->d_past || !st->cos_out || !st->sin_out || st->vocab <= 0) return P3V_ERR_ARG;
    if (((uintptr_t)st->embed_table | (uintptr_t)st->x_out) & 15) return P3V_ERR_ARG;
  } else {
    if (!a->x || !st->tok_out || !st->history || !st->d_step || !st->d_past || !st->ticket || !st->amax_ws) return P3V_ERR_ARG;
    if ((uintptr_t)st->amax_ws & 7) return P3V_ERR_ARG;
  }
  const GemvF8P p = {a->x, a->W, a->w_scale, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue, a->N / 2};
  const GemvStepP sp = {st->tok, st->embed_table, st->vocab, st->x_out, st->cos_t, st->sin_t, st->d_past, st->cos_out, st->sin_out, st->tab_t,
                        st->half_dim, st->next_tok, st->tok_out, st->history, st->d_step, st->d_past, st->ticket, st->amax_ws, st->max_steps};
  hipStream_t s = (hipStream_t)stream;
  if (a->K == 3072) return begin ? launch_gemv3_f8<1, 3, STEP_BEGIN>(p, s, &sp) : launch_gemv3_f8<1, 3, STEP_END>(p, s, &sp);
  return begin ? launch_gemv3_f8<2, 4, STEP_BEGIN>(p, s, &sp) : launch_gemv3_f8<2, 4, STEP_END>(p, s, &sp);
}

extern "C" int p3v_gemv_fp8(const p3v_gemv_fp8_args_t* a, void* stream) {
  if (!a || !a->x || !a->W || !a->w_scale || !a->out) return P3V_ERR_ARG;
  if (a->M <= 0 || a->M > 16 || a->N <= 0 || a->N % 2 || (a->K != 3072 && a->K != 8192)) return P3V_ERR_UNSUPPORTED;
  if (a->epilogue != P3V_EPI_NONE && a->epilogue != P3V_EPI_RESID_BF16 && a->epilogue != P3V_EPI_SILU_MUL &&
      a->epilogue != P3V_EPI_F32)
    return P3V_ERR_UNSUPPORTED;
  if (a->epilogue == P3V_EPI_RESID_BF16 && !a->resid) return P3V_ERR_ARG;
  GemvF8P p = {a->x, a->W, a->w_scale, a->out, a->resid, a->norm_w, a->norm_eps, a->M, a->N, a->K, a->epilogue,
               a->epilogue == P3V_EPI_SILU_MUL ? a->N : a->N / 2};
  hipStream_t s = (hipStream_t)stream;
  if (a->M == 1) return a->K == 3072 ? launch_gemv3_f8<1, 3>(p, s) : launch_gemv3_f8<2, 4>(p, s);
  if (a->M >= 5 && a->M <= 8 && !p3v_tuning().gemv_no_mfma8) {
    const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
    if (a->K == 3072) return silu ? launch_gemv_mfma8_f8<true, 4, 6>(p, s) : launch_gemv_mfma8_f8<false, 4, 6>(p, s);
    return silu ? launch_gemv_mfma8_f8<true, 8, 8>(p, s) : launch_gemv_mfma8_f8<false, 8, 8>(p, s);
  }
  dim3 grid(p3v_cdiv(a->N, 16));
  if (a->epilogue == P3V_EPI_SILU_MUL) hipLaunchKernelGGL(k_gemv_mfma_f8<true>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(k_gemv_mfma_f8<false>, grid, dim3(256), 0, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
