// BASELINE config 5, prefill: W8A8 projection on the gfx950 fp8 matrix cores.
//
//   C[M,N] = (sa[m] * sw[n]) * sum_k A8[m,k] * W8[n,k]        A8, W8 = OCP e4m3 bytes, K contiguous
//
// replaces the reference's QuantizedLinear call sites (phi_3_vision_mlx.py:264, 291-305) for prompt-sized inputs:
// weights are the per-output-row-scaled e4m3 bytes of `quantize_model=True` (the same bytes the decode GEMV streams),
// activations are quantised on the fly, one fp32 scale per token row (p3v_quant_fp8_rows, fused with the RMSNorm that
// precedes qkv_proj / gate_up_proj).
//
//   MFMA      v_mfma_scale_f32_16x16x128_f8f6f4 with both block scales = 1.0 (E8M0 0x7f): the only fp8 form that runs at
//             twice the bf16 rate on this chip (the unscaled 16x16x32 fp8 MFMA runs at the bf16 rate).  A lane holds 32
//             k bytes of its row (lane & 15) for k-block lane >> 4 -- the SAME bytes for A and B, so which k they are
//             does not matter (the kernel takes 16-byte chunks fchunk and fchunk + 4: bank-conflict-free reads).
//   tile      256(M) x 256(N) x 128(K) per K step, 512 threads = 8 waves as 2(M) x 4(N), wave tile 128 x 64 = 8 x 4
//             accumulators; ONE MFMA per accumulator and K step.  The LDS image is the bf16 kernel's (p3v_gemm256.hip):
//             rows of 128 BYTES, 16-byte chunk c of row r stored at chunk c ^ (r & 7), staged by LDS-DMA with the swizzle
//             on the source address, two 64-KiB buffers, half-tiles of K-step t+1 requested in the first two phases of t.
//             Same bytes per K step as the bf16 kernel, twice the k: the step costs the same, the flops double.
//   epilogue  scales, then none / residual add / SiLU(gate) * up, bf16 out (wave-private LDS tile -> 16-byte stores).
#include <stdlib.h>

#include "p3v_common.h"

#define TM 256
#define TN 256
#define TKB 128                        // K bytes (= fp8 elements) per step
#define HALF_BYTES (128 * TKB)         // 16 KiB: 128 rows x 128 B
#define BUF_BYTES (4 * HALF_BYTES)     // A0 A1 B0 B1
#define CT_LD 68
#define GEMMF8_LDS (2 * BUF_BYTES)     // 128 KiB

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

struct GemmF8P {
  const uint8_t* A; const float* sa; const uint8_t* W; const float* sw; bf16_t* out; const bf16_t* resid;
  int M, N, K, lda, ldw, ldo;
};

// NJ = 16-column blocks per wave: 4 -> 256 x 256 tile; 2 -> 256 x 128 tile (one B half-tile, 96 KiB of LDS in use): the
// N = 3072 projections (o_proj, down_proj) have only 120 tiles of 256 x 256 for 256 CUs -- 240 narrower tiles fill the chip.
template <int EPI, int NJ = 4>
__global__ void __launch_bounds__(512, 1) k_gemm256_f8(GemmF8P p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  static_assert(NJ == 4 || (NJ == 2 && !SILU), "narrow tiles: plain epilogues only");
  constexpr int TNW = NJ * 64;                       // tile width in W rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  constexpr int n_out_tile = SILU ? TNW / 2 : TNW;
  int m_t, n_t;
  {   // XCD-aware tile order (as p3v_gemm256.hip): each XCD gets a contiguous run of tiles, bands of 4 M tiles
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy, wid = blockIdx.y * gx + blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    constexpr int BAND = 4;
    const int band = id / (BAND * gx), in_band = id % (BAND * gx);
    const int rows = min(BAND, gy - band * BAND);
    m_t = band * BAND + in_band % rows;
    n_t = in_band / rows;
  }
  const int m0 = m_t * TM, n0 = n_t * n_out_tile;

  const int srow = tid >> 3, schunk = tid & 7;
  int a_off[2][2], b_off[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = h * 128 + q * 64 + srow;
      const int sw = (schunk ^ (r & 7)) * 16;
      const int ar = min(m0 + r, p.M - 1) - m0;
      a_off[h][q] = ar * p.lda + sw;
      int br;
      if (SILU) {                                   // wave column group wcol (64 tile rows) = 32 gate + 32 up rows
        const int wcol = r >> 6, ni = (r & 63) >> 4, c = r & 15;
        br = min(n0 + wcol * 32 + (ni & 1) * 16 + c, p.N - 1) + (ni >> 1) * p.N;
      } else {
        br = min(n0 + r, p.N - 1);
      }
      b_off[h][q] = (int)((unsigned)br * (unsigned)p.ldw + (unsigned)sw);
    }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)m0 * p.lda), 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
  auto dma_half = [&](int which, int kt, int buf) {   // which: 0 A0, 1 A1, 2 B0, 3 B1
    unsigned char* base = smem + buf * BUF_BYTES + which * HALF_BYTES + wave * 1024;
    const int h = which & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(which < 2 ? rs_a : rs_w, (lptr_t)(base + q * 8192), 16,
                                               which < 2 ? a_off[h][q] : b_off[h][q], kt * TKB, 0, 0);
  };

  f32x4_t acc[8][NJ];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fchunk = lane >> 4;
  const int nk = p.K / TKB;
#pragma unroll
  for (int w4 = 0; w4 < (NJ == 4 ? 4 : 3); ++w4) dma_half(w4, 0, 0);

  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = kt + 1 < nk;
    const int nb = (kt + 1) & 1;
    const unsigned char* ta = smem + (kt & 1) * BUF_BYTES + wr * HALF_BYTES;
    const unsigned char* tb = smem + (kt & 1) * BUF_BYTES + (2 + (wc >> 1)) * HALF_BYTES + (wc & 1) * 64 * 128;
    if (NJ == 2) tb = smem + (kt & 1) * BUF_BYTES + 2 * HALF_BYTES + wc * 32 * 128;                   // 32 B rows per wave
    u32x4_t af[4][2], af1[4][2], bf0[2][2], bf1[2][2];
    // lane (row frow, k-block fchunk) owns the 16-byte chunks fchunk and fchunk + 4 of its 128-byte row (A and B alike, so the
    // k permutation cancels).  Contiguous chunks 2*fchunk, 2*fchunk + 1 -- the obvious choice -- put the 16 lanes of a
    // ds_read_b128 group on 8 bank slots, two-way conflicts on every fragment read (PMC: SQ_LDS_BANK_CONFLICT 48 % of the
    // LDS-active cycles); with chunks (kk * 4 + fchunk) the group covers all 16 slots, as in the bf16 kernel
    auto read_a_to = [&](int sub, u32x4_t (&dst)[4][2]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int r = sub * 64 + i * 16 + frow;
          dst[i][kk] = *(const u32x4_t*)(ta + r * 128 + (((fchunk + 4 * kk) ^ (r & 7)) << 4));
        }
    };
    auto read_b = [&](int sub, u32x4_t (&bf)[2][2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int r = sub * 32 + j * 16 + frow;
          bf[j][kk] = *(const u32x4_t*)(tb + r * 128 + (((fchunk + 4 * kk) ^ (r & 7)) << 4));
        }
    };
    auto quad_from = [&](int asub, int bsub, u32x4_t (&a)[4][2], u32x4_t (&bf)[2][2]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const i32x8_t av = {(int)a[i][0][0], (int)a[i][0][1], (int)a[i][0][2], (int)a[i][0][3],
                              (int)a[i][1][0], (int)a[i][1][1], (int)a[i][1][2], (int)a[i][1][3]};
          const i32x8_t bv = {(int)bf[j][0][0], (int)bf[j][0][1], (int)bf[j][0][2], (int)bf[j][0][3],
                              (int)bf[j][1][0], (int)bf[j][1][1], (int)bf[j][1][2], (int)bf[j][1][3]};
          acc[asub * 4 + i][bsub * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              av, bv, acc[asub * 4 + i][bsub * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      __builtin_amdgcn_s_setprio(0);
    };
    auto dma_phase = [&](int ph) {                    // half-tiles A0 A1 in phase 0, B0 B1 in phase 1 (SCHED 0x50 of the bf16 kernel)
      if (more) {
        if (ph == 0) { dma_half(0, kt + 1, nb); dma_half(1, kt + 1, nb); }
        else { dma_half(2, kt + 1, nb); if (NJ == 4) dma_half(3, kt + 1, nb); }
      }
    };
    if (NJ == 4) {
      read_b(0, bf0);
      read_a_to(0, af);
      dma_phase(0);
      read_b(1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      quad_from(0, 0, af, bf0);
      __builtin_amdgcn_sched_barrier(0);
      read_a_to(1, af1);
      dma_phase(1);
      __builtin_amdgcn_sched_barrier(0);
      quad_from(0, 1, af, bf1);
      __builtin_amdgcn_sched_barrier(0);
      quad_from(1, 1, af1, bf1);
      quad_from(1, 0, af1, bf0);
    } else {
      read_b(0, bf0);
      read_a_to(0, af);
      dma_phase(0);
      read_a_to(1, af1);
      dma_phase(1);
      __builtin_amdgcn_sched_barrier(0);
      quad_from(0, 0, af, bf0);
      quad_from(1, 0, af1, bf0);
    }
  }

  // ---- epilogue: 128 x 64 per wave, four passes of 32 rows through a wave-private [32][68] fp32 LDS tile
  __syncthreads();
  float* ct = (float*)smem + wave * (32 * CT_LD);
  const int ccol = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[(i * 16 + crow + r) * CT_LD + j * 16 + ccol] = acc[pass * 2 + i][j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (SILU) {
      const int c8 = (lane & 3) * 8, n = n0 + wc * 32 + c8;
      float sg[8], su[8];
      if (n < p.N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg[e] = p.sw[n + e]; su[e] = p.sw[p.N + n + e]; }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2);
        const int m = m0 + wr * 128 + pass * 32 + row;
        if (m < p.M && n < p.N) {
          const float s = p.sa[m];
          const float4 g0 = *(const float4*)(ct + row * CT_LD + c8), g1 = *(const float4*)(ct + row * CT_LD + c8 + 4);
          const float4 u0 = *(const float4*)(ct + row * CT_LD + 32 + c8), u1 = *(const float4*)(ct + row * CT_LD + 36 + c8);
          const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          const float us[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
          u32x4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float o2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              // gate / up are bf16 Linear outputs, every elementwise op after them rounds to bf16 (phi.py:469-471)
              const float g = bf16_round(gs[2 * e + h] * (s * sg[2 * e + h])), u = bf16_round(us[2 * e + h] * (s * su[2 * e + h]));
              o2[h] = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * u;
            }
            w[e] = pack_bf16x2(o2[0], o2[1]);
          }
          *(u32x4_t*)(p.out + (size_t)m * p.ldo + n) = w;
        }
      }
    } else {
      // a row of the wave tile is NJ * 16 columns = NJ * 2 lanes of 8 columns: 8 (4) rows per pass of the 64 lanes
      constexpr int LPR = NJ * 2, RPI = 64 / LPR;
      const int c8 = (lane % LPR) * 8, n = n0 + wc * (NJ * 16) + c8;
      const bool ncol_ok = n < p.N;
      float sc[8];
      if (ncol_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sc[e] = p.sw[n + e];
      }
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + lane / LPR;
        const int m = m0 + wr * 128 + pass * 32 + row;
        if (m < p.M && ncol_ok) {
          const float s = p.sa[m];
          const float4 a0 = *(const float4*)(ct + row * CT_LD + c8), a1 = *(const float4*)(ct + row * CT_LD + c8 + 4);
          float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= s * sc[e];
          const size_t o = (size_t)m * p.ldo + n;
          if (EPI == P3V_EPI_RESID_BF16) {
            const u32x4_t rw = *(const u32x4_t*)(p.resid + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo(rw[j]) + bf16_round(v[2 * j]); v[2 * j + 1] = bf16hi(rw[j]) + bf16_round(v[2 * j + 1]); }
          }
          u32x4_t w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
          *(u32x4_t*)(p.out + o) = w;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int EPI, int NJ>
static int launch_gemm_f8_v(const GemmF8P& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm256_f8<EPI, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMMF8_LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const int n_tile = EPI == P3V_EPI_SILU_MUL ? TN / 2 : NJ * 64;
  dim3 grid(p3v_cdiv(p.N, n_tile), p3v_cdiv(p.M, TM));
  hipLaunchKernelGGL((k_gemm256_f8<EPI, NJ>), grid, dim3(512), GEMMF8_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// tile width: the wide tile unless it leaves most of the chip idle (fewer wide tiles than ~0.6 x 256 CUs, or a last round
// that narrow tiles fill better); P3V_GEMM_F8_NARROW=0/1 pins the choice (kernel tests run both)
template <int EPI>
static int launch_gemm_f8(const GemmF8P& p, hipStream_t s) {
  if constexpr (EPI == P3V_EPI_SILU_MUL) {
    return launch_gemm_f8_v<EPI, 4>(p, s);
  } else {
    const long mt = p3v_cdiv(p.M, TM), wide = mt * p3v_cdiv(p.N, 256), narrow = mt * p3v_cdiv(p.N, 128);
    bool use_narrow = p.N % 256 != 0 || wide <= 160 || (wide % 256 != 0 && wide % 256 <= 64 && narrow % 256 > 128);
    if (const int f = p3v_tuning().gemm_f8_narrow; f >= 0) use_narrow = p.N % 256 != 0 || f != 0;
    return use_narrow ? launch_gemm_f8_v<EPI, 2>(p, s) : launch_gemm_f8_v<EPI, 4>(p, s);
  }
}

extern "C" int p3v_gemm_fp8(const p3v_gemm_fp8_args_t* a, void* stream) {
  if (!a || !a->A || !a->a_scale || !a->W || !a->w_scale || !a->out) return P3V_ERR_ARG;
  const int n_tile = 128;                              // SILU: 128 outputs per wide tile; plain: the narrow tile
  if (a->M < 0 || a->N <= 0 || a->K <= 0 || a->K % TKB || a->N % n_tile || a->ldo % 8) return P3V_ERR_ARG;
  if (a->lda < a->K || a->ldw < a->K || a->lda % 16 || a->ldw % 16) return P3V_ERR_ARG;
  if (((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->out | (uintptr_t)a->resid) & 15) return P3V_ERR_ARG;
  if (a->epilogue == P3V_EPI_RESID_BF16 && !a->resid) return P3V_ERR_ARG;
  const size_t w_rows = (size_t)a->N * (a->epilogue == P3V_EPI_SILU_MUL ? 2 : 1);
  if (w_rows * a->ldw >= ((size_t)1 << 32) || (size_t)256 * a->lda >= ((size_t)1 << 31)) return P3V_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  if (a->M == 0) return P3V_OK;
  const GemmF8P p = {a->A, a->a_scale, a->W, a->w_scale, a->out, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo};
  hipStream_t s = (hipStream_t)stream;
  switch (a->epilogue) {
    case P3V_EPI_NONE: return launch_gemm_f8<P3V_EPI_NONE>(p, s);
    case P3V_EPI_RESID_BF16: return launch_gemm_f8<P3V_EPI_RESID_BF16>(p, s);
    case P3V_EPI_SILU_MUL: return launch_gemm_f8<P3V_EPI_SILU_MUL>(p, s);
    default: return P3V_ERR_ARG;
  }
}

// ---------------------------------------------------------------- activation quantiser (one wave per token row)
// q[m, k] = e4m3(h[m, k] * (1 / s[m])),  s[m] = max_k |h[m, k]| / 448 (1 for an all-zero row; 1 / s rounded to fp32 once per
// row -- 8 million IEEE divisions per call kept the first version VALU-bound),  h = x, or -- with a norm weight --
// h = bf16(x * rsqrt(mean x^2 + eps) * g): exactly the row p3v_rmsnorm would have written (phi.py:478-479, 482, 484).
// The row is read ONCE and stays in registers (CH 16-byte chunks per lane: 6 for K = 3072, 16 for K = 8192) through the
// sum of squares, the maximum and the conversion: 3 bytes of traffic per element.
template <bool NORM, int CH>
__global__ void __launch_bounds__(256) k_quant_fp8_rows(const u32x4_t* __restrict__ x, const u32x4_t* __restrict__ g,
                                                        u32x2_t* __restrict__ q, float* __restrict__ scale, int rows,
                                                        int chunks, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const u32x4_t* xr = x + (size_t)row * chunks;
  u32x4_t v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < chunks ? xr[c] : (u32x4_t){0u, 0u, 0u, 0u};
  }
  float r = 1.f;
  if (NORM) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16lo(v[i][j]), b = bf16hi(v[i][j]);
        ss += a * a + b * b;
      }
    r = rsqrtf(wave_sum(ss) * inv_h + eps);
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    if (NORM && c < chunks) {
      const u32x4_t w = g[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = rms_pair(v[i][j], r, w[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(v[i][j])), fabsf(bf16hi(v[i][j]))));
  }
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax / 448.f : 1.f;
  const float inv = 1.f / s;
  if (lane == 0) scale[row] = s;
  u32x2_t* qr = q + (size_t)row * chunks;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 64 + lane;
    if (c < chunks) {
      u32x2_t o;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[i][2 * j]) * inv, bf16hi(v[i][2 * j]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[i][2 * j + 1]) * inv, bf16hi(v[i][2 * j + 1]) * inv, w, true);
        o[j] = (uint32_t)w;
      }
      qr[c] = o;
    }
  }
}

template <bool NORM, int CH>
static void launch_quant(const uint16_t* x, const uint16_t* norm_w, float eps, uint8_t* q, float* scale, int rows, int K, hipStream_t s) {
  hipLaunchKernelGGL((k_quant_fp8_rows<NORM, CH>), dim3(p3v_cdiv(rows, 4)), dim3(256), 0, s, (const u32x4_t*)x, (const u32x4_t*)norm_w,
                     (u32x2_t*)q, scale, rows, K / 8, 1.0f / K, eps);
}

extern "C" int p3v_quant_fp8_rows(const uint16_t* x, const uint16_t* norm_w, float eps, uint8_t* q, float* scale, int rows,
                                  int K, void* stream) {
  if (!x || !q || !scale || rows < 0 || K <= 0 || K % 8 || K > 16 * 64 * 8) return P3V_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)norm_w) & 15 || (uintptr_t)q & 7) return P3V_ERR_ARG;
  if (rows == 0) return P3V_OK;
  hipStream_t s = (hipStream_t)stream;
  const int ch = p3v_cdiv(K / 8, 64);
  if (norm_w) {
    if (ch <= 6) launch_quant<true, 6>(x, norm_w, eps, q, scale, rows, K, s);
    else launch_quant<true, 16>(x, norm_w, eps, q, scale, rows, K, s);
  } else {
    if (ch <= 6) launch_quant<false, 6>(x, norm_w, eps, q, scale, rows, K, s);
    else launch_quant<false, 16>(x, norm_w, eps, q, scale, rows, K, s);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
