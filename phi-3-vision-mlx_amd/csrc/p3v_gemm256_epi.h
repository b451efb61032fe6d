// Shared by the two 256 x 256-tile GEMM kernels (p3v_gemm256.hip, p3v_gemm256pp.hip): launch parameters and the epilogue.
#pragma once
#include "p3v_gemm_qkv.h"

#define P3V_EPI_QKV 100                // internal: the qkv projection with split + RoPE + KV append in the epilogue (p3v_gemm_qkv.h)

#define TM 256
#define TN 256
#define TK 64
#define HALF_BYTES (128 * TK * 2)     // 16 KiB: 128 rows of one operand's K-tile

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Gemm256P {
  const bf16_t* A; const bf16_t* W; void* out; const bf16_t* bias; const void* resid;
  int M, N, K, lda, ldw, ldo;
  QkvP q;                                // P3V_EPI_QKV only
};

__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// ---- epilogue: 128 x 64 per wave, STRAIGHT FROM THE ACCUMULATORS (round 5).  The MFMAs take the W fragment as their
// first operand, so a 16 x 16 block comes out transposed: lane (c = lane & 15, q = lane >> 4) holds, for output row
// m = block row c, the FOUR CONSECUTIVE columns 4q .. 4q+3 (the plain operand order gives four rows of one column, which is
// why rounds 1-4 staged every tile through a wave-private LDS image: 64 ds_write_b32 + 32 ds_read_b128 per lane and tile,
// ~4 us per tile).  fp32 outputs: those four values are one 16-byte store.  bf16 outputs: the lane pairs (q, q ^ 1) trade the
// packed halves of two neighbouring column blocks with v_permlane16_swap, after which every lane owns 8 consecutive columns
// = one 16-byte store (even q: block 2jp, odd q: block 2jp + 1; columns 8 * (q >> 1) .. + 8 of it).  No LDS, no barrier:
// a wave starts its stores as soon as ITS last MFMA is done.
// acc[i][j]: block row i (16 output rows), block column j (16 W rows) of the wave's 128 x 64 sub-tile at (wr, wc) of the tile at (m0, n0).
template <int EPI>
__device__ __forceinline__ void gemm256_epilogue(const Gemm256P& p, f32x4_t (&acc)[8][4], int m0, int n0, int wr, int wc, int lane) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  const int fc = lane & 15, fq = lane >> 4;
  const int mrow0 = m0 + wr * 128 + fc;                                  // + i * 16
#if defined(P3V_G256_ABL) && P3V_G256_ABL == 3                  // timing experiment: no epilogue at all
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
#else
  if (SILU) {
    // wave columns: blocks 0, 1 = gate, blocks 2, 3 = up of the SAME 32 output columns n0 + wc*32 + [0, 32)
    const int n = n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t pk[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // reference rounds gate/up to bf16 (Linear output) and every elementwise op after it (phi.py:469-471)
          const float g = bf16_round(acc[i][j][r]), u = bf16_round(acc[i][2 + j][r]);
          o4[r] = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * u;
        }
        pk[j][0] = pack_bf16x2(o4[0], o4[1]), pk[j][1] = pack_bf16x2(o4[2], o4[3]);
      }
      u32x4_t w;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        auto sw = __builtin_amdgcn_permlane16_swap(pk[0][k], pk[1][k], false, false);
        w[k] = sw[0], w[2 + k] = sw[1];
      }
      const int m = mrow0 + i * 16;
      if (m < p.M && n < p.N) *(u32x4_t*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = w;
    }
  } else if (EPI == P3V_EPI_BIAS_RESID_F32 || EPI == P3V_EPI_F32) {
    float bias[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fq * 4;
      u32x2_t bw = {0u, 0u};
      if (p.bias && n < p.N) bw = *(const u32x2_t*)(p.bias + n);
      bias[j][0] = bf16lo(bw[0]), bias[j][1] = bf16hi(bw[0]), bias[j][2] = bf16lo(bw[1]), bias[j][3] = bf16hi(bw[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mrow0 + i * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fq * 4;
        if (m < p.M && n < p.N) {
          const size_t o = (size_t)m * p.ldo + n;
          float4 v = make_float4(acc[i][j][0] + bias[j][0], acc[i][j][1] + bias[j][1], acc[i][j][2] + bias[j][2], acc[i][j][3] + bias[j][3]);
          if (EPI == P3V_EPI_BIAS_RESID_F32) {
            const float4 r4 = *(const float4*)((const float*)p.resid + o);
            v.x += r4.x, v.y += r4.y, v.z += r4.z, v.w += r4.w;
          }
          *(float4*)((float*)p.out + o) = v;
        }
      }
    }
  } else {
    float bias[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fq * 4;
      u32x2_t bw = {0u, 0u};
      if (p.bias && n < p.N) bw = *(const u32x2_t*)(p.bias + n);
      bias[j][0] = bf16lo(bw[0]), bias[j][1] = bf16hi(bw[0]), bias[j][2] = bf16lo(bw[1]), bias[j][3] = bf16hi(bw[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mrow0 + i * 16;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        uint32_t pk[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] + bias[j][r];
            if (EPI == P3V_EPI_BIAS_QGELU) v[r] = v[r] * p3v_sigmoid(1.702f * v[r]);
            else if (EPI == P3V_EPI_BIAS_GELU) v[r] = gelu_erf2(v[r]);
          }
          pk[jj][0] = pack_bf16x2(v[0], v[1]), pk[jj][1] = pack_bf16x2(v[2], v[3]);
        }
        u32x4_t w;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          auto sw = __builtin_amdgcn_permlane16_swap(pk[0][k], pk[1][k], false, false);
          w[k] = sw[0], w[2 + k] = sw[1];
        }
        const int n = n0 + wc * 64 + (2 * jp + (fq & 1)) * 16 + (fq >> 1) * 8;
        if (m < p.M && n < p.N) {
          const size_t o = (size_t)m * p.ldo + n;
          if (EPI == P3V_EPI_RESID_BF16) {                     // out = resid + bf16(acc): the packed words ARE bf16(acc)
            const u32x4_t rw = *(const u32x4_t*)((const bf16_t*)p.resid + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(bf16lo(rw[k]) + bf16lo(w[k]), bf16hi(rw[k]) + bf16hi(w[k]));
          }
#if defined(P3V_G256_ABL) && P3V_G256_ABL == 1                  // timing experiment: everything but the global store
          asm volatile("" :: "v"(w));
#else
          *(u32x4_t*)((bf16_t*)p.out + o) = w;
#endif
        }
      }
    }
  }
#endif
}
