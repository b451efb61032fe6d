// Dense projection  C[M,N] = A[M,K] * W[N,K]^T  on the gfx950 matrix cores.
//
// Both operands are K-contiguous, which is exactly the MFMA A/B fragment order
// (each lane owns 8 consecutive k), so no transposes anywhere.
//
//   tile      128(M) x 128(N) x 64(K), 256 threads = 4 waves as 2(M) x 2(N),
//             each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators
//   staging   global_load_lds_dwordx4 (LDS-DMA, 1 KiB = 8 rows x 128 B per
//             wave-instruction), two LDS buffers (64 KiB), one barrier per K tile
//   LDS       rows of 128 B; 16-B chunk c of row r is stored at chunk c^(r&7)
//             (XOR swizzle applied on the per-lane SOURCE address because the
//             DMA destination is lane-linear; the same XOR on the ds_read_b128
//             side) -> conflict-free fragment reads
//   epilogue  runtime switch (bias / quick-GELU / erf-GELU / residual / SiLU*up /
//             patch-embed row remap), fp32 math, bf16 or fp32 stores
//
// Roofline: MFMA-bound for M >= ~512 (2*M*N*K flops vs (M+N)*K*2 bytes).
#include "p3v_common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)   // 16 KiB per operand tile

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmP {
  const bf16_t* A; const bf16_t* W; void* out; const bf16_t* bias; const void* resid; const bf16_t* pos;
  int M, N, K, lda, ldw, ldo, epi, ppi, n_wrows;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <bool SILU>
__global__ void __launch_bounds__(256, 2) k_gemm(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int n_out_tile = SILU ? BN / 2 : BN;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * n_out_tile;

  // ---- staging addresses: wave w issues 4 DMA pieces per operand, piece q covers tile rows (w*4+q)*8 .. +8
  const int srow = lane >> 3, schunk = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (wave * 4 + q) * 8 + srow;                 // tile row 0..127
    const int sw = (schunk ^ (r & 7)) * 8;                   // swizzled source chunk (elements)
    int ar = m0 + r;
    ar = ar < p.M ? ar : p.M - 1;
    a_src[q] = p.A + (size_t)ar * p.lda + sw;
    int br;
    if (SILU) {
      const int wcol = r >> 6, ni = (r & 63) >> 4, c = r & 15;
      br = n0 + wcol * 32 + (ni & 1) * 16 + c;
      br = (br < p.N ? br : p.N - 1) + (ni >> 1) * p.N;      // up rows live N rows below the gate rows
    } else {
      br = n0 + r;
      br = br < p.N ? br : p.N - 1;
    }
    b_src[q] = p.W + (size_t)br * p.ldw + sw;
  }
  const int nk = p.K / BK;

  auto stage = [&](int kt, int buf) {
    unsigned char* base = smem + buf * (2 * TILE_BYTES);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = (wave * 4 + q) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[q] + (size_t)kt * BK), (lptr_t)(base + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[q] + (size_t)kt * BK), (lptr_t)(base + TILE_BYTES + piece), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a tile: row = base + (lane&15), logical chunk = kk*4 + (lane>>4)
  const int frow = lane & 15, fchunk = lane >> 4;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const unsigned char* ta = smem + (kt & 1) * (2 * TILE_BYTES);
    const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + frow;
        af[i] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + frow;
        bfr[j] = *(const bf16x8_t*)(tb + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue.  C layout of the 16x16 tile: col = lane&15, row = (lane>>4)*4 + r
  const int ccol = lane & 15, crow = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * 64 + i * 16 + crow + r;
      if (m >= p.M) continue;
      if (SILU) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wc * 32 + j * 16 + ccol;
          if (n >= p.N) continue;
          // reference rounds gate/up to bf16 (Linear output) before silu*up (phi.py:469-471)
          const float g = bf16_round(acc[i][j][r]), u = bf16_round(acc[i][j + 2][r]);
          const float s = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g))));
          ((bf16_t*)p.out)[(size_t)m * p.ldo + n] = f32_to_bf16(s * u);
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + ccol;
        if (n >= p.N) continue;
        float v = acc[i][j][r];
        if (p.bias) v += bf16_to_f32(p.bias[n]);
        const size_t o = (size_t)m * p.ldo + n;
        switch (p.epi) {
          case P3V_EPI_NONE:
          case P3V_EPI_BIAS: ((bf16_t*)p.out)[o] = f32_to_bf16(v); break;
          case P3V_EPI_BIAS_QGELU: ((bf16_t*)p.out)[o] = f32_to_bf16(v / (1.f + __expf(-1.702f * v))); break;
          case P3V_EPI_BIAS_GELU: ((bf16_t*)p.out)[o] = f32_to_bf16(gelu_erf(v)); break;
          case P3V_EPI_BIAS_RESID_F32: ((float*)p.out)[o] = ((const float*)p.resid)[o] + v; break;
          case P3V_EPI_RESID_BF16:
            ((bf16_t*)p.out)[o] = f32_to_bf16(bf16_to_f32(((const bf16_t*)p.resid)[o]) + bf16_round(v));
            break;
          case P3V_EPI_F32: ((float*)p.out)[o] = v; break;
          case P3V_EPI_PATCH: {
            const int img = m / p.ppi, pi = m % p.ppi;
            ((float*)p.out)[((size_t)img * (p.ppi + 1) + 1 + pi) * p.ldo + n] =
                v + bf16_to_f32(p.pos[(size_t)(1 + pi) * p.N + n]);
          } break;
        }
      }
    }
  }
}

extern "C" int p3v_gemm(const p3v_gemm_args_t* a, void* stream) {
  if (!a || !a->A || !a->W || !a->out) return P3V_ERR_ARG;
  if (a->M < 0 || a->N <= 0 || a->K <= 0 || a->K % BK) return P3V_ERR_ARG;
  if (a->lda < a->K || a->ldw < a->K || a->lda % 8 || a->ldw % 8) return P3V_ERR_ARG;
  if (((uintptr_t)a->A | (uintptr_t)a->W) & 15) return P3V_ERR_ARG;
  if ((a->epilogue == P3V_EPI_BIAS_RESID_F32 || a->epilogue == P3V_EPI_RESID_BF16) && !a->resid) return P3V_ERR_ARG;
  if (a->epilogue == P3V_EPI_PATCH && (!a->pos || a->patches_per_img <= 0)) return P3V_ERR_ARG;
  if (a->epilogue < 0 || a->epilogue > P3V_EPI_F32) return P3V_ERR_ARG;
  if (a->M == 0) return P3V_OK;
  GemmP p = {a->A, a->W, a->out, a->bias, a->resid, a->pos, a->M, a->N, a->K, a->lda, a->ldw, a->ldo,
             a->epilogue, a->patches_per_img, 0};
  const size_t lds = 4 * TILE_BYTES;
  if (a->epilogue == P3V_EPI_SILU_MUL) {
    dim3 grid(p3v_cdiv(a->N, BN / 2), p3v_cdiv(a->M, BM));
    hipLaunchKernelGGL(k_gemm<true>, grid, dim3(256), lds, (hipStream_t)stream, p);
  } else {
    dim3 grid(p3v_cdiv(a->N, BN), p3v_cdiv(a->M, BM));
    hipLaunchKernelGGL(k_gemm<false>, grid, dim3(256), lds, (hipStream_t)stream, p);
  }
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
