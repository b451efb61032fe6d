// The qkv projection with the head split, the rotation and the KV append in its epilogue (round 5): p3v_gemm_qkv =
// p3v_gemm(qkv_proj) + p3v_rope_kv_append in the launches of the GEMM alone.  Shared by the 256 x 256 and the 128 x 128 tile kernels.
//
// What makes it fit the accumulator layout (a lane holds 4 consecutive output columns of one row, p3v_gemm256_epi.h):
//   * Q and K tiles: _rotate_half pairs dimension d with d + hd/2 (phi.py:419-423).  The tile's W rows are fetched in PAIR order --
//     a wave's 64 tile columns are 32 consecutive pairs: blocks 0, 1 their first halves, blocks 2, 3 their second halves (the DMA takes
//     any W row for any tile row: the same trick as the SiLU epilogue's gate / up rows) -- so both halves of a pair sit in ONE lane
//     (acc[i][j][r] and acc[i][j + 2][r]) and the rotation is lane-local.  16 pairs never straddle a head (hd/2 = 48 or 32).
//   * V tiles: the cache keeps V TRANSPOSED ([.., hd, t]).  These tiles are computed with the operand roles swapped -- W rows fill the
//     tile's A side, the tokens its B side (the tile is square and the K loop does not care) -- so a lane holds 4 consecutive TOKENS of
//     one V dimension, and 8 after the v_permlane16_swap: one 16-byte store into a V^T row (rows whose length is not a multiple of 8
//     -- CLIP's 577-token crops -- shift the run inside the row: still one dwordx4 store at even offsets, three stores at odd ones).
// Arithmetic: exactly p3v_gemm's + p3v_rope_kv_append's (the Linear output rounded to bf16, rotation in fp32 on those values,
// q_scale before the one rounding): bit-identical to the two launches (tests/test_kernels_gpu.py::test_gemm_qkv_fused).
#pragma once
#include "p3v_common.h"

struct QkvP {
  const float* cos_t; const float* sin_t;        // [B / tab_div, tab_t, hd / 2] or both null: plain head split (CLIP)
  bf16_t* q_out; bf16_t* k_dst; bf16_t* v_dst;   // [B, nh, L, hd], [B, nkv, dst_t, hd], [B, nkv, hd, dst_t]
  int L, nh, nkv, hd, past, dpos0, dst_t, tab_t, tab_div;
  int m_base;                                    // global token index of this launch's row 0 (rows after a big / small split)
  float q_scale;
};

// W row fetched for tile column `rr` (0 .. tile width - 1) of a Q / K tile whose first pair is p0; row0 = first W row of the region
__device__ __forceinline__ int qkv_pair_row(const QkvP& q, int row0, int p0, int rr) {
  const int half = q.hd >> 1;
  const int wcol = rr >> 6, ni = (rr & 63) >> 4, c = rr & 15;
  const int P = p0 + wcol * 32 + (ni & 1) * 16 + c;
  return row0 + (P / half) * q.hd + P % half + (ni >> 1) * half;
}

// Q / K tile: acc[i][j] = block row i (16 tokens from m_first), blocks 0, 1 = first halves of pairs pw .. pw + 31, blocks 2, 3 = second
template <int NI>
__device__ __forceinline__ void qkv_epilogue_rot(const QkvP& q, f32x4_t (&acc)[NI][4], const bf16_t* bias, bool is_k, int row0, int pw,
                                                 int m_first, int M, int lane) {
  const int fc = lane & 15, fq = lane >> 4, half = q.hd >> 1;
  const bool rot = q.cos_t != nullptr;
  const float qs = is_k ? 1.f : q.q_scale;
  int head[2], d0[2];
  float b1[2][4], b2[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int P = pw + j * 16 + fq * 4;
    head[j] = P / half, d0[j] = P % half;
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[j][r] = b2[j][r] = 0.f;
    if (bias) {
      const u32x2_t w1 = *(const u32x2_t*)(bias + row0 + head[j] * q.hd + d0[j]);
      const u32x2_t w2 = *(const u32x2_t*)(bias + row0 + head[j] * q.hd + d0[j] + half);
      b1[j][0] = bf16lo(w1[0]), b1[j][1] = bf16hi(w1[0]), b1[j][2] = bf16lo(w1[1]), b1[j][3] = bf16hi(w1[1]);
      b2[j][0] = bf16lo(w2[0]), b2[j][1] = bf16hi(w2[0]), b2[j][2] = bf16lo(w2[1]), b2[j][3] = bf16hi(w2[1]);
    }
  }
  // the block this lane stores after the swap: jj = fq & 1, its columns 8 * (fq >> 1) .. + 8
  const int Ps = pw + (fq & 1) * 16, head_s = Ps / half, d_s = Ps % half + (fq >> 1) * 8;
  // (b, l) of the lane's row in block 0; later blocks are 16 tokens further on (one division per tile, not per block)
  const int m00 = m_first + fc;
  int b0 = 0, l0 = q.m_base + (m00 < M ? m00 : M - 1);
  if (l0 >= q.L) { b0 = l0 / q.L; l0 -= b0 * q.L; }
  // The rotation tables cost 4 sixteen-byte loads per block row: requested FOUR BLOCK ROWS AT A TIME (64 registers; the fragment
  // registers of the K loop are free here), so a tile pays two round trips to the L2 instead of eight dependent ones.
  constexpr int G = NI < 4 ? NI : 4;
#pragma unroll
  for (int ig = 0; ig < NI; ig += G) {
    float4 cs4[G][2], sn4[G][2];
    int bb[G], ll[G];
#pragma unroll
    for (int ii = 0; ii < G; ++ii) {
      const int i = ig + ii;
      int l = l0 + i * 16, b = b0;
      const int m = m00 + i * 16;
      if (m >= M) l = l0 + (M - 1 - (m00 < M ? m00 : M - 1));   // clamped rows: any valid table row (nothing is stored for them)
      while (l >= q.L) { l -= q.L; ++b; }
      bb[ii] = b, ll[ii] = l;
      if (rot) {
        const size_t trow = ((size_t)(b / q.tab_div) * q.tab_t + q.past + l) * half;
#pragma unroll
        for (int j = 0; j < 2; ++j) { cs4[ii][j] = *(const float4*)(q.cos_t + trow + d0[j]); sn4[ii][j] = *(const float4*)(q.sin_t + trow + d0[j]); }
      }
    }
#pragma unroll
    for (int ii = 0; ii < G; ++ii) {
      const int i = ig + ii, m = m00 + i * 16, b = bb[ii], l = ll[ii];
      uint32_t pk1[2][2], pk2[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float cs[4] = {cs4[ii][j].x, cs4[ii][j].y, cs4[ii][j].z, cs4[ii][j].w}, sn[4] = {sn4[ii][j].x, sn4[ii][j].y, sn4[ii][j].z, sn4[ii][j].w};
        float o1[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a0 = bf16_round(acc[i][j][r] + b1[j][r]), a1 = bf16_round(acc[i][j + 2][r] + b2[j][r]);   // the Linear's bf16 output
          if (rot) {
            p3v_rope_pair(a0, a1, cs[r], sn[r], qs, o1[r], o2[r]);
          } else {
            o1[r] = qs != 1.f ? a0 * qs : a0;
            o2[r] = qs != 1.f ? a1 * qs : a1;
          }
        }
        pk1[j][0] = pack_bf16x2(o1[0], o1[1]), pk1[j][1] = pack_bf16x2(o1[2], o1[3]);
        pk2[j][0] = pack_bf16x2(o2[0], o2[1]), pk2[j][1] = pack_bf16x2(o2[2], o2[3]);
      }
      u32x4_t w1, w2;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        auto s1 = __builtin_amdgcn_permlane16_swap(pk1[0][k], pk1[1][k], false, false);
        w1[k] = s1[0], w1[2 + k] = s1[1];
        auto s2 = __builtin_amdgcn_permlane16_swap(pk2[0][k], pk2[1][k], false, false);
        w2[k] = s2[0], w2[2 + k] = s2[1];
      }
      if (m < M) {
        bf16_t* dst = is_k ? q.k_dst + (((size_t)b * q.nkv + head_s) * q.dst_t + q.dpos0 + l) * q.hd
                           : q.q_out + (((size_t)b * q.nh + head_s) * q.L + l) * q.hd;
        *(u32x4_t*)(dst + d_s) = w1;
        *(u32x4_t*)(dst + d_s + half) = w2;
      }
    }
  }
}

// V tile (operand roles swapped): acc[i][j] = block row i of 16 V dimensions from fv_first (index into [nkv * hd]), block column j of
// 16 tokens from t_first.  bias_v = bias + first W row of the V region, or null.
template <int NI>
__device__ __forceinline__ void qkv_epilogue_vt(const QkvP& q, f32x4_t (&acc)[NI][4], const bf16_t* bias_v, int fv_first, int t_first,
                                                int M, int lane) {
  const int fc = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int fv = fv_first + i * 16 + fc, head = fv / q.hd, d = fv - head * q.hd;
    const float bv = bias_v ? bf16_to_f32(bias_v[fv]) : 0.f;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint32_t pk[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * jp + jj;
        pk[jj][0] = pack_bf16x2(acc[i][j][0] + bv, acc[i][j][1] + bv), pk[jj][1] = pack_bf16x2(acc[i][j][2] + bv, acc[i][j][3] + bv);
      }
      u32x4_t w;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        auto sw = __builtin_amdgcn_permlane16_swap(pk[0][k], pk[1][k], false, false);
        w[k] = sw[0], w[2 + k] = sw[1];
      }
      const int t = t_first + (2 * jp + (fq & 1)) * 16 + (fq >> 1) * 8;      // 8 consecutive tokens t .. t + 7
      if (t < M) {
        const int mg = q.m_base + t, b = mg < q.L ? 0 : mg / q.L, l = mg - b * q.L;
        bf16_t* dst = q.v_dst + (((size_t)b * q.nkv + head) * q.hd + d) * (size_t)q.dst_t + q.dpos0 + l;
        if (t + 8 <= M && l + 8 <= q.L) {                      // the run lies in one batch row
          if (!(l & 1)) {
            // even token offset: the run is dword-aligned (16-byte aligned for one batch row, or rows of L % 8 == 0 tokens; round 6:
            // any row length -- CLIP's 577-token crops start one token later in every crop) -- one dwordx4 store either way
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            *(u32x4_a4*)dst = w;
          } else {                                             // odd offset: first token, three dwords of token pairs, last token
            typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
            dst[0] = (bf16_t)w[0];
            const u32x3_a4 mid = {__builtin_amdgcn_alignbyte(w[1], w[0], 2), __builtin_amdgcn_alignbyte(w[2], w[1], 2),
                                  __builtin_amdgcn_alignbyte(w[3], w[2], 2)};
            *(u32x3_a4*)(dst + 1) = mid;
            dst[7] = (bf16_t)(w[3] >> 16);
          }
        } else {                                               // the ragged end of the prompt, or a run across two batch rows
          for (int e = 0; e < 8 && t + e < M; ++e) {
            int le = l + e, be = b;
            if (le >= q.L) { le -= q.L; ++be; }
            q.v_dst[(((size_t)be * q.nkv + head) * q.hd + d) * (size_t)q.dst_t + q.dpos0 + le] = (bf16_t)(w[e >> 1] >> (16 * (e & 1)));
          }
        }
      }
    }
  }
}
