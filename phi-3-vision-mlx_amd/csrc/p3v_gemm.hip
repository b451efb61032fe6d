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
//   epilogue  compile-time variant (bias / quick-GELU / erf-GELU / residual /
//             SiLU*up / patch-embed row remap), straight from the accumulators: the
//             MFMAs take the W fragment FIRST, so a lane holds 4 consecutive columns
//             of one output row; bf16 outputs pair two column blocks with
//             v_permlane16_swap -> 16-byte stores of 8 consecutive columns, no LDS
//             staging (rounds 1-4 went through a wave-private LDS tile).
//
// Roofline: MFMA-bound for M >= ~512 (2*M*N*K flops vs (M+N)*K*2 bytes).
#include <stdlib.h>
#include <string.h>


#include "p3v_gemm_qkv.h"

#define P3V_EPI_QKV 100                // internal: qkv projection with split + RoPE + KV append in the epilogue (p3v_gemm_qkv.h)

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)   // 16 KiB per operand tile
#define GEMM_LDS (4 * TILE_BYTES)   // two stages of an A and a W tile: 64 KiB -> two workgroups per CU

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct GemmP {
  const bf16_t* A; const bf16_t* W; void* out; const bf16_t* bias; const void* resid; const bf16_t* pos;
  int M, N, K, lda, ldw, ldo, ppi;
  int kslice;   // split-K (EPI_F32 only): workgroup z covers k in [z*kslice, (z+1)*kslice) and writes fp32 partial z ([M, ldo] each)
  QkvP q;       // P3V_EPI_QKV only
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ void load8_bf16(const bf16_t* p, float* v) {
  const u32x4_t w = *(const u32x4_t*)p;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = bf16lo(w[j]); v[2 * j + 1] = bf16hi(w[j]); }
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float* v) {
  u32x4_t w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
  *(u32x4_t*)p = w;
}

template <int EPI, int ORD = 0>
__global__ void __launch_bounds__(256, 2) k_gemm(GemmP p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL, QKV = EPI == P3V_EPI_QKV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: LDS-DMA bases stay in SGPRs
  const int wr = wave >> 1, wc = wave & 1;
  const int n_out_tile = SILU ? BN / 2 : BN;
  // XCD-aware tile order: the dispatcher places workgroup i on XCD i % 8 (speed only, never correctness), so
  // give each XCD a contiguous run of tiles; inside a run tiles walk N fastest within a band of 8 M-tiles,
  // i.e. the tiles resident on one XCD share A row panels and W panels through that XCD's 4 MiB L2.
  int m_t, n_t;
  {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy, wid = blockIdx.y * gx + blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = wid & 7, loc = wid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;       // bijective for any nwg
    constexpr int BAND = 8;
    const int band = id / (BAND * gx), in_band = id % (BAND * gx);
    const int rows = min(BAND, gy - band * BAND);
    m_t = band * BAND + in_band % rows;
    n_t = in_band / rows;
  }
  const int m0 = m_t * BM;
  int n0 = n_t * n_out_tile;
  // P3V_EPI_QKV (p3v_gemm_qkv.h): column tiles 0 .. nq-1 are Q, then K (both fetched in rotation-pair order), then V, whose tiles
  // are computed with the operand roles swapped (W rows on the tile's A side, tokens on its B side)
  bool swapped = false, is_k = false;
  int row0 = 0;
  if (QKV) {
    const int nq_t = p.q.nh * p.q.hd / BN, nk_t = p.q.nkv * p.q.hd / BN;
    swapped = n_t >= nq_t + nk_t;
    is_k = !swapped && n_t >= nq_t;
    row0 = swapped ? (p.q.nh + p.q.nkv) * p.q.hd : is_k ? p.q.nh * p.q.hd : 0;
    n0 = (n_t - (swapped ? nq_t + nk_t : is_k ? nq_t : 0)) * BN;       // first column INSIDE the region
  }

  // ---- staging addresses: wave w issues 4 DMA pieces per operand, piece q covers tile rows (w*4+q)*8 .. +8
  const int srow = lane >> 3, schunk = lane & 7;
  // byte offsets from p.A / p.W: the DMA is issued in its MUBUF form (`buffer_load ... lds`).  A pending `global_load_lds`
  // is, for the compiler's wait-count model, a FLAT access of both memories, and every later s_waitcnt becomes
  // vmcnt(0) / lgkmcnt(0): the counted lgkmcnt on the fragment reads below was silently a full wait.
  unsigned a_src[4], b_src[4];
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ww = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = QKV && swapped ? rs_ww : rs_x, rs_w = QKV && swapped ? rs_x : rs_ww;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (wave * 4 + q) * 8 + srow;                 // tile row 0..127
    const int sw = (schunk ^ (r & 7)) * 8;                   // swizzled source chunk (elements)
    int ar = m0 + r;
    ar = ar < p.M ? ar : p.M - 1;
    int br;
    if (SILU) {
      const int wcol = r >> 6, ni = (r & 63) >> 4, c = r & 15;
      br = n0 + wcol * 32 + (ni & 1) * 16 + c;
      br = (br < p.N ? br : p.N - 1) + (ni >> 1) * p.N;      // up rows live N rows below the gate rows
    } else if (QKV) {
      br = qkv_pair_row(p.q, row0, n0 >> 1, r);
    } else {
      br = n0 + r;
      br = br < p.N ? br : p.N - 1;
    }
    if (QKV && swapped) {                                    // A side: W rows row0 + n0 + r; B side: token rows (clamped)
      a_src[q] = (unsigned)(((size_t)(row0 + n0 + r) * p.ldw + sw) * 2);
      b_src[q] = (unsigned)(((size_t)ar * p.lda + sw) * 2);
    } else {
      a_src[q] = (unsigned)(((size_t)ar * p.lda + sw) * 2);
      b_src[q] = (unsigned)(((size_t)br * p.ldw + sw) * 2);
    }
  }
  const int kz = p.kslice ? (int)blockIdx.z : 0;              // split-K slice (0 when not split)
  const int nk = (p.kslice ? p.kslice : p.K) / BK, kt0 = kz * nk;

  auto stage = [&](int kt, int buf) {
    unsigned char* base = smem + buf * (2 * TILE_BYTES);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = (wave * 4 + q) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(base + piece), 16, a_src[q], (kt0 + kt) * (BK * 2), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(base + TILE_BYTES + piece), 16, b_src[q], (kt0 + kt) * (BK * 2), 0, 0);
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
    if (ORD == 0 && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const unsigned char* ta = smem + (kt & 1) * (2 * TILE_BYTES);
    const unsigned char* tb = ta + TILE_BYTES;
    // All 16 fragment reads of the K-tile are requested up front (64 VGPRs) and the MFMAs of the first k-step start as
    // soon as ITS eight have landed (LDS returns in order -> counted lgkmcnt): one exposed LDS latency per K-tile instead
    // of four (hipcc otherwise recycles two A registers and waits lgkmcnt(0) before every run of eight MFMAs).
    bf16x8_t af[2][4], bfr[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + frow;
        af[kk][i] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + frow;
        bfr[kk][j] = *(const bf16x8_t*)(tb + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
    }
    if (ORD == 1 && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);   // W first: transposed block
      if (ORD == 2 && kk == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue straight from the accumulators (round 5; rounds 1-4 staged the tile through LDS between two barriers).  The
  // MFMAs take the W fragment first, so a 16 x 16 block comes out transposed: lane (c = lane & 15, q = lane >> 4) holds the four
  // consecutive columns 4q .. 4q+3 of output row c.  fp32 outputs store them as they are (16 bytes); bf16 outputs first trade
  // packed halves of two neighbouring column blocks between lanes q and q ^ 1 (v_permlane16_swap): 8 consecutive columns per lane.
  if constexpr (QKV) {
    if (!swapped) qkv_epilogue_rot<4>(p.q, acc, p.bias, is_k, row0, (n0 >> 1) + wc * 32, m0 + wr * 64, p.M, lane);
    else qkv_epilogue_vt<4>(p.q, acc, p.bias ? p.bias + row0 : nullptr, n0 + wr * 64, m0 + wc * 64, p.M, lane);
    return;
  }
  const int fc = lane & 15, fq = lane >> 4;
  const int mrow0 = m0 + wr * 64 + fc;                         // + i * 16
  if (SILU) {
    // wave tile columns: [0,32) gate, [32,64) up for output columns n0 + wc*32 + [0,32)
    const int n = n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
    return;
  }

  float bias[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + fq * 4;
    u32x2_t bw = {0u, 0u};
    if (p.bias && n < p.N) bw = *(const u32x2_t*)(p.bias + n);
    bias[j][0] = bf16lo(bw[0]), bias[j][1] = bf16hi(bw[0]), bias[j][2] = bf16lo(bw[1]), bias[j][3] = bf16hi(bw[1]);
  }
  if (EPI == P3V_EPI_BIAS_RESID_F32 || EPI == P3V_EPI_F32 || EPI == P3V_EPI_PATCH) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mrow0 + i * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fq * 4;
        if (m < p.M && n < p.N) {
          const size_t o = (size_t)m * p.ldo + n;
          size_t oo = o;
          float4 v = make_float4(acc[i][j][0] + bias[j][0], acc[i][j][1] + bias[j][1], acc[i][j][2] + bias[j][2], acc[i][j][3] + bias[j][3]);
          if (EPI == P3V_EPI_BIAS_RESID_F32) {
            const float4 r4 = *(const float4*)((const float*)p.resid + o);
            v.x += r4.x, v.y += r4.y, v.z += r4.z, v.w += r4.w;
          } else if (EPI == P3V_EPI_PATCH) {
            const int img = m / p.ppi, pi = m % p.ppi;
            const u32x2_t pw = *(const u32x2_t*)(p.pos + (size_t)(1 + pi) * p.N + n);
            v.x += bf16lo(pw[0]), v.y += bf16hi(pw[0]), v.z += bf16lo(pw[1]), v.w += bf16hi(pw[1]);
            oo = ((size_t)img * (p.ppi + 1) + 1 + pi) * p.ldo + n;
          }
          if (EPI == P3V_EPI_F32) oo += (size_t)kz * p.M * p.ldo;
          *(float4*)((float*)p.out + oo) = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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
          else if (EPI == P3V_EPI_BIAS_GELU) v[r] = gelu_erf(v[r]);
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
        if (EPI == P3V_EPI_RESID_BF16) {                       // out = resid + bf16(acc): the packed words ARE bf16(acc)
          const u32x4_t rw = *(const u32x4_t*)((const bf16_t*)p.resid + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(bf16lo(rw[k]) + bf16lo(w[k]), bf16hi(rw[k]) + bf16hi(w[k]));
        }
        *(u32x4_t*)((bf16_t*)p.out + o) = w;
      }
    }
  }
}

template <int EPI, int ORD>
static int launch_gemm_v(const GemmP& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm<EPI, ORD>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const int n_tile = EPI == P3V_EPI_SILU_MUL ? BN / 2 : BN;
  dim3 grid(p3v_cdiv(p.N, n_tile), p3v_cdiv(p.M, BM), p.kslice ? p.K / p.kslice : 1);
  hipLaunchKernelGGL((k_gemm<EPI, ORD>), grid, dim3(256), GEMM_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <int EPI>
static int launch_gemm(const GemmP& p, hipStream_t s) { return launch_gemm_v<EPI, 0>(p, s); }

int p3v_gemm256_try(const p3v_gemm_args_t* a, hipStream_t s);   // p3v_gemm256.hip: 256x256 tiles, P3V_ERR_UNSUPPORTED if the shape does not fit
static int gemm128(const p3v_gemm_args_t* a, hipStream_t s);

// How many leading rows go to the 256x256 kernel (a multiple of 256, or M for all of them); the rest runs on the
// 128x128 kernel in a second launch on the same stream.  The machine holds 256 big workgroups or 512 small ones
// per round; a round of big tiles costs 1, a round of small ones 0.675 (half the output area at 0.97 vs 1.31
// PFLOP/s), a last round that leaves every CU at most one small workgroup 0.42 -- pick the cheapest whole-round
// packing, e.g. 2531 x 16384 (gate_up): 8 x 64 = 512 big tiles = 2 rounds + 4 x 128 small = 1 round.
static int gemm_big_rows(const p3v_gemm_args_t* a) {
  const int n_big = a->epilogue == P3V_EPI_SILU_MUL ? 128 : 256, n_small = n_big / 2;
  if (a->M < 1024 || a->N % n_big || a->epilogue == P3V_EPI_PATCH) return 0;
  if (const int force = p3v_tuning().gemm_big_rows; force >= 0) return min(force / 256 * 256, a->M);   // tests: pin the split
  const int mt = p3v_cdiv(a->M, 256), nt_big = a->N / n_big, nt_small = p3v_cdiv(a->N, n_small);
  // a short K loop leaves the big tile's 128-KiB prologue and four-pass epilogue exposed (one workgroup per CU)
  const float big_round = a->K >= 2048 ? 1.0f : 1.0f + 0.25f * (2048 - a->K) / 1024.f;
  float best = 1e30f;
  int best_rows = 0;
  for (int ms = 0; ms <= mt; ++ms) {
    const int rows_big = min(ms * 256, a->M), rows_small = a->M - rows_big;
    const long big = (long)ms * nt_big, small = (long)p3v_cdiv(rows_small, 128) * nt_small;
    float c = (float)((big + 255) / 256) * big_round;
    if (small) {
      const long full = small / 512, rest = small % 512;
      c += 0.675f * full + (rest == 0 ? 0.f : rest <= 256 ? 0.42f : 0.675f);
    }
    if (big && small) c += 0.04f;                       // the second launch
    if (c < best - 1e-4f) { best = c; best_rows = rows_big; }
  }
  return best_rows;
}

// ---- split-K for prompt-sized-but-small M (17 .. 256 rows: short chat prompts, the text group of a mixed batch).  One
// 128-row tile leaves N / 128 = 24 .. 128 workgroups, each walking the WHOLE K loop with one tile in flight: an iteration is
// one DMA round trip (~0.6 us), so down_proj (K = 8192) took 78 us for a 50 MB weight matrix and a 17..256-token prefill
// 8.3-9.5 ms, flat in the length.  Here S slices of K run as gridDim.z (fp32 partials [S, M, N] in the CALLER's workspace:
// p3v_gemm_ws_bytes() sizes it, the library never allocates) and a second launch adds them in slice order (deterministic)
// and applies the epilogue.  Without a workspace of that size the shape runs on the one-pass kernels.
template <int EPI>
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ part, void* __restrict__ out, const void* __restrict__ resid,
                                                       int M, int N, int ldp, int ldo, int S) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;             // part holds [gate | up] columns (2N), out N columns
  const int per_row = N / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * per_row) return;
  const int m = (int)(i / per_row), n = (int)(i - (long)m * per_row) * 8;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int z = 0; z < S; ++z) {
    const float* pr = part + ((size_t)z * M + m) * ldp + n;
    const float4 a0 = *(const float4*)pr, a1 = *(const float4*)(pr + 4);
    v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    if (SILU) {
      const float4 b0 = *(const float4*)(pr + N), b1 = *(const float4*)(pr + N + 4);
      u[0] += b0.x; u[1] += b0.y; u[2] += b0.z; u[3] += b0.w; u[4] += b1.x; u[5] += b1.y; u[6] += b1.z; u[7] += b1.w;
    }
  }
  const size_t o = (size_t)m * ldo + n;
  if (SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = bf16_round(v[e]), up = bf16_round(u[e]);
      v[e] = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * up;
    }
  } else if (EPI == P3V_EPI_RESID_BF16) {
    float r[8];
    load8_bf16((const bf16_t*)resid + o, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r[e] + bf16_round(v[e]);
  }
  store8_bf16((bf16_t*)out + o, v);
}

// The reduction of a split whose consumer is an RMSNorm (o_proj / down_proj of a short prompt): adds the S partials in slice order,
// out = resid + bf16(sum) exactly as k_splitk_reduce<P3V_EPI_RESID_BF16>, and writes RMSNorm(out) beside it -- the residual stream
// and the next projection's input from ONE launch instead of two.  A workgroup per row, a thread per 8 columns (every load of the
// launch in flight at once: with k_rmsnorm_r's one wave per row, 128 rows keep 128 waves busy and the launch takes longer than the
// two it replaces).  The sum of squares is k_rmsnorm_r's, bit for bit: that kernel's lane l adds p = lo*lo + fl(hi*hi) of its chunks
// l, l + 64, ... one pair at a time; here every thread computes the four p of ITS chunk, and wave 0 adds them in that order.
__global__ void __launch_bounds__(384) k_splitk_reduce_norm(const float* __restrict__ part, bf16_t* __restrict__ out, const bf16_t* __restrict__ resid,
                                                            const u32x4_t* __restrict__ norm_w, u32x4_t* __restrict__ normed, int M, int N,
                                                            int ldo, int S, float inv_h, float eps) {
  __shared__ float ps[6 * 4][64];
  __shared__ float r_sh;
  const int row = blockIdx.x, c = threadIdx.x, lane = c & 63, wv = c >> 6, chunks = N / 8;
  u32x4_t v = {0u, 0u, 0u, 0u}, g = {0u, 0u, 0u, 0u};
  if (c < chunks) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int z = 0; z < S; ++z) {
      const float* pr = part + ((size_t)z * M + row) * N + c * 8;
      const float4 a0 = *(const float4*)pr, a1 = *(const float4*)(pr + 4);
      acc[0] += a0.x; acc[1] += a0.y; acc[2] += a0.z; acc[3] += a0.w; acc[4] += a1.x; acc[5] += a1.y; acc[6] += a1.z; acc[7] += a1.w;
    }
    const size_t o = (size_t)row * ldo + c * 8;
    float r[8];
    load8_bf16(resid + o, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = r[e] + bf16_round(acc[e]);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(acc[2 * j], acc[2 * j + 1]);
    *(u32x4_t*)(out + o) = v;
    g = norm_w[c];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = bf16lo(v[j]), hi = bf16hi(v[j]);
    float hh = hi * hi;
    asm volatile("" : "+v"(hh));                                // (one rounding of hi*hi, then ONE fma: k_rmsnorm_r's compiled form)
    ps[wv * 4 + j][lane] = __builtin_fmaf(lo, lo, hh);
  }
  __syncthreads();
  if (wv == 0) {
    float ss = ps[0][lane];
#pragma unroll
    for (int k = 1; k < 24; ++k) {
      ss += ps[k][lane];
      asm volatile("" : "+v"(ss));
    }
    const float r = rsqrtf(wave_sum(ss) * inv_h + eps);
    if (lane == 0) r_sh = r;
  }
  __syncthreads();
  if (c < chunks) {
    const float r = r_sh;
    u32x4_t o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = rms_pair(v[j], r, g[j]);
    normed[(size_t)row * chunks + c] = o4;
  }
}

// number of K slices for a shape (1 = not a split-K shape)
static int splitk_slices(int M, int N, int K, int epilogue) {
  const P3vTuning& t = p3v_tuning();
  const bool silu = epilogue == P3V_EPI_SILU_MUL;
  if (t.gemm_no_splitk || M <= 16 || M > t.gemm_splitk_max_m /* beyond, every projection has >= 256 tiles anyway */ ||
      N % 128 || K % 512)
    return 1;
  if (epilogue != P3V_EPI_NONE && epilogue != P3V_EPI_RESID_BF16 && !silu) return 1;
  const int w_rows = silu ? 2 * N : N;                        // SiLU: [gate; up] taken as 2N plain output columns
  const int tiles = p3v_cdiv(M, BM) * (w_rows / BN);
  int S = 1;
  while (S < t.gemm_splitk_max_s && tiles * S < t.gemm_splitk_wgs && (K / (2 * S)) % BK == 0 && K / (2 * S) >= 2 * BK) S *= 2;
  return S;
}

int p3v_gemm_skinny_slices(int M, int N, int K, int epilogue);   // p3v_gemm_skinny.hip: 0 = not one of its shapes
int p3v_gemm_skinny_try(const p3v_gemm_args_t* a, hipStream_t s);

extern "C" int64_t p3v_gemm_ws_bytes(int M, int N, int K, int epilogue) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int sk = p3v_gemm_skinny_slices(M, N, K, epilogue);
  const int S = sk ? sk : splitk_slices(M, N, K, epilogue);
  if (S == 1) return 0;
  return (int64_t)S * M * (epilogue == P3V_EPI_SILU_MUL ? 2 * N : N) * 4;
}

// ---- out = resid + bf16(A W^T) and normed = RMSNorm(out) (include/p3v.h): only where the projection runs as K slices anyway
int p3v_gemm_skinny_partials(const p3v_gemm_args_t* a, int* S_out, hipStream_t s);   // p3v_gemm_skinny.hip
extern "C" int p3v_gemm_resid_norm(const p3v_gemm_args_t* a, const uint16_t* norm_w, float eps, uint16_t* normed, void* stream) {
  if (!a || !a->A || !a->W || !a->out || !a->resid || !norm_w || !normed) return P3V_ERR_ARG;
  if (a->epilogue != P3V_EPI_RESID_BF16 || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->K % BK || a->N % 8 || a->ldo % 8 || a->ldo < a->N) return P3V_ERR_ARG;
  if (a->lda < a->K || a->ldw < a->K || a->lda % 8 || a->ldw % 8) return P3V_ERR_ARG;
  if (((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->out | (uintptr_t)a->resid | (uintptr_t)norm_w | (uintptr_t)normed | (uintptr_t)a->ws) & 15)
    return P3V_ERR_ARG;
  if (a->N > 6 * 64 * 8 || a->bias || (size_t)a->N * a->ldw * 2 >= (1ull << 32)) return P3V_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int S = 0;
  const int rc = p3v_gemm_skinny_partials(a, &S, s);           // P3V_ERR_UNSUPPORTED (nothing launched) unless the shape splits
  if (rc != P3V_OK) return rc;
  hipLaunchKernelGGL(k_splitk_reduce_norm, dim3(a->M), dim3(384), 0, s, (const float*)a->ws, (bf16_t*)a->out, (const bf16_t*)a->resid,
                     (const u32x4_t*)norm_w, (u32x4_t*)normed, a->M, a->N, a->ldo, S, 1.0f / (float)a->N, eps);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// the second launch of a split: adds the S fp32 partials [S, M, W rows] in slice order and applies the epilogue
int p3v_splitk_reduce(const float* part, const p3v_gemm_args_t* a, int S, hipStream_t s) {
  const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
  const int w_rows = silu ? 2 * a->N : a->N;
  const long items = (long)a->M * (a->N / 8);
  const dim3 grid((unsigned)p3v_cdiv(items, 256));
  if (silu) hipLaunchKernelGGL(k_splitk_reduce<P3V_EPI_SILU_MUL>, grid, dim3(256), 0, s, part, a->out, a->resid, a->M, a->N, w_rows, a->ldo, S);
  else if (a->epilogue == P3V_EPI_RESID_BF16)
    hipLaunchKernelGGL(k_splitk_reduce<P3V_EPI_RESID_BF16>, grid, dim3(256), 0, s, part, a->out, a->resid, a->M, a->N, w_rows, a->ldo, S);
  else hipLaunchKernelGGL(k_splitk_reduce<P3V_EPI_NONE>, grid, dim3(256), 0, s, part, a->out, a->resid, a->M, a->N, w_rows, a->ldo, S);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

// returns P3V_ERR_UNSUPPORTED when the shape is not one for split-K (or the caller gave no workspace for it)
static int gemm_splitk(const p3v_gemm_args_t* a, hipStream_t s) {
  const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
  const int S = splitk_slices(a->M, a->N, a->K, a->epilogue);
  if (S == 1) return P3V_ERR_UNSUPPORTED;
  const int w_rows = silu ? 2 * a->N : a->N;
  if (!a->ws || a->ws_bytes < (int64_t)S * a->M * w_rows * 4) return P3V_ERR_UNSUPPORTED;
  if ((uintptr_t)a->ws & 15) return P3V_ERR_ARG;
  float* part = (float*)a->ws;
  GemmP p = {a->A, a->W, part, nullptr, nullptr, nullptr, a->M, w_rows, a->K, a->lda, a->ldw, w_rows, 0, a->K / S, {}};
  const int rc = launch_gemm<P3V_EPI_F32>(p, s);
  if (rc != P3V_OK) return rc;
  return p3v_splitk_reduce(part, a, S, s);
}

extern "C" int p3v_gemm(const p3v_gemm_args_t* a, void* stream) {
  if (!a || !a->A || !a->W || !a->out) return P3V_ERR_ARG;
  if (a->M < 0 || a->N <= 0 || a->K <= 0 || a->K % BK || a->N % 8 || a->ldo % 8) return P3V_ERR_ARG;
  if (a->lda < a->K || a->ldw < a->K || a->lda % 8 || a->ldw % 8) return P3V_ERR_ARG;
  if (((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->out | (uintptr_t)a->resid | (uintptr_t)a->bias | (uintptr_t)a->pos) & 15)
    return P3V_ERR_ARG;
  if ((a->epilogue == P3V_EPI_BIAS_RESID_F32 || a->epilogue == P3V_EPI_RESID_BF16) && !a->resid) return P3V_ERR_ARG;
  if (a->epilogue == P3V_EPI_PATCH && (!a->pos || a->patches_per_img <= 0)) return P3V_ERR_ARG;
  if (a->epilogue < 0 || a->epilogue > P3V_EPI_F32) return P3V_ERR_ARG;
  if (a->M == 0) return P3V_OK;
  {                                                            // operands are addressed with 32-bit byte offsets (buffer loads)
    const size_t w_rows = (size_t)a->N * (a->epilogue == P3V_EPI_SILU_MUL ? 2 : 1);
    if ((size_t)a->M * a->lda * 2 >= (1ull << 32) || w_rows * a->ldw * 2 >= (1ull << 32)) return P3V_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!a->bias) {                                              // 17 .. 256 rows: the weight-streaming kernel (p3v_gemm_skinny.hip)
    const int rc = p3v_gemm_skinny_try(a, s);
    if (rc != P3V_ERR_UNSUPPORTED) return rc;
  }
  {
    const int rc = gemm_splitk(a, s);
    if (rc != P3V_ERR_UNSUPPORTED) return rc;
  }
  const bool big_tiles = !p3v_tuning().gemm_128;
  const int rows_big = big_tiles ? gemm_big_rows(a) : 0;
  if (rows_big > 0) {
    p3v_gemm_args_t top = *a;
    top.M = rows_big;
    const int rc = p3v_gemm256_try(&top, s);
    if (rc == P3V_OK && rows_big < a->M) {
      const bool f32_out = a->epilogue == P3V_EPI_BIAS_RESID_F32 || a->epilogue == P3V_EPI_F32;
      const size_t out_row = (size_t)a->ldo * (f32_out ? 4 : 2);
      p3v_gemm_args_t rest = *a;
      rest.M = a->M - rows_big;
      rest.A = a->A + (size_t)rows_big * a->lda;
      rest.out = (char*)a->out + rows_big * out_row;
      if (a->resid) rest.resid = (const char*)a->resid + rows_big * out_row;
      return gemm128(&rest, s);
    }
    if (rc != P3V_ERR_UNSUPPORTED) return rc;
  }
  return gemm128(a, s);
}

static int gemm128(const p3v_gemm_args_t* a, hipStream_t s) {
  const GemmP p = {a->A, a->W, a->out, a->bias, a->resid, a->pos,
                   a->M, a->N, a->K, a->lda, a->ldw, a->ldo, a->patches_per_img, 0, {}};
  switch (a->epilogue) {
    case P3V_EPI_NONE: return launch_gemm<P3V_EPI_NONE>(p, s);
    case P3V_EPI_BIAS: return launch_gemm<P3V_EPI_BIAS>(p, s);
    case P3V_EPI_BIAS_QGELU: return launch_gemm<P3V_EPI_BIAS_QGELU>(p, s);
    case P3V_EPI_BIAS_GELU: return launch_gemm<P3V_EPI_BIAS_GELU>(p, s);
    case P3V_EPI_BIAS_RESID_F32: return launch_gemm<P3V_EPI_BIAS_RESID_F32>(p, s);
    case P3V_EPI_RESID_BF16: return launch_gemm<P3V_EPI_RESID_BF16>(p, s);
    case P3V_EPI_SILU_MUL: return launch_gemm<P3V_EPI_SILU_MUL>(p, s);
    case P3V_EPI_PATCH: return launch_gemm<P3V_EPI_PATCH>(p, s);
    case P3V_EPI_F32: return launch_gemm<P3V_EPI_F32>(p, s);
    default: return P3V_ERR_ARG;
  }
}

// ---- the qkv projection with the head split, the rotation and the KV append in its epilogue (p3v_gemm_qkv.h)
int p3v_gemm256_qkv(const p3v_gemm_args_t* a, const QkvP& q, hipStream_t s);   // p3v_gemm256.hip
int p3v_gemm_skinny_qkv(const p3v_gemm_args_t* a, const QkvP& q, hipStream_t s);   // p3v_gemm_skinny.hip

extern "C" int p3v_gemm_qkv(const p3v_gemm_args_t* a, const p3v_qkv_split_t* sp, void* stream) {
  if (!a || !sp || !a->A || !a->W || !sp->q_out || !sp->k_dst || !sp->v_dst || (!sp->cos_t) != (!sp->sin_t)) return P3V_ERR_ARG;
  if (a->epilogue != P3V_EPI_NONE && a->epilogue != P3V_EPI_BIAS) return P3V_ERR_ARG;
  if (a->epilogue == P3V_EPI_BIAS && !a->bias) return P3V_ERR_ARG;
  const int nh = sp->n_heads, nkv = sp->n_kv, hd = sp->hd, half = hd / 2;
  if (sp->B <= 0 || sp->L <= 0 || nh <= 0 || nkv <= 0 || hd <= 0 || sp->tab_div <= 0 || !(sp->q_scale > 0.f)) return P3V_ERR_ARG;
  if (a->M != sp->B * sp->L || a->N != (nh + 2 * nkv) * hd || a->K % BK || a->lda < a->K || a->ldw < a->K || a->lda % 8 || a->ldw % 8) return P3V_ERR_ARG;
  if (((uintptr_t)a->A | (uintptr_t)a->W | (uintptr_t)a->bias | (uintptr_t)sp->q_out | (uintptr_t)sp->k_dst | (uintptr_t)sp->v_dst |
       (uintptr_t)sp->cos_t | (uintptr_t)sp->sin_t) & 15)
    return P3V_ERR_ARG;
  // what the fused epilogue needs (anything else: P3V_ERR_UNSUPPORTED, the caller runs p3v_gemm + p3v_rope_kv_append):
  // prompt-sized M; rotation pairs in blocks of 16 inside a head; whole 128-column tiles per region; an 8-aligned append offset
  const int dpos0 = sp->dst_off_is_past ? sp->past : 0;
  // (round 6: batch rows of any length -- the V^T runs are then stored at whatever offset they fall on, qkv_epilogue_vt)
  if (hd % 32 || half % 16 || dpos0 % 8 || sp->dst_t % 8 || p3v_tuning().gemm_no_qkv_fuse) return P3V_ERR_UNSUPPORTED;
  if ((size_t)a->M * a->lda * 2 >= (1ull << 32) || (size_t)a->N * a->ldw * 2 >= (1ull << 32)) return P3V_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  QkvP q = {sp->cos_t, sp->sin_t, sp->q_out, sp->k_dst, sp->v_dst, sp->L, nh, nkv, hd, sp->past, dpos0, sp->dst_t, sp->tab_t, sp->tab_div, 0,
            sp->q_scale};
  if (a->M < 1024) return p3v_gemm_skinny_qkv(a, q, s);       // 17 .. 256 rows: the weight-streaming kernel (or "unsupported")
  if ((nh * hd) % BN || (nkv * hd) % BN || p3v_tuning().gemm_128) return P3V_ERR_UNSUPPORTED;
  p3v_gemm_args_t plain = *a;
  plain.epilogue = P3V_EPI_NONE;                                // (the row packing below prices the plain bf16 epilogue)
  int rows_big = (nh * hd) % 256 || (nkv * hd) % 256 ? 0 : gemm_big_rows(&plain);
  if (rows_big > 0) {
    p3v_gemm_args_t top = *a;
    top.M = rows_big;
    const int rc = p3v_gemm256_qkv(&top, q, s);
    if (rc == P3V_ERR_UNSUPPORTED) rows_big = 0;
    else if (rc != P3V_OK) return rc;
  }
  if (rows_big < a->M) {
    q.m_base = rows_big;
    const GemmP p = {a->A + (size_t)rows_big * a->lda, a->W, nullptr, a->bias, nullptr, nullptr, a->M - rows_big, a->N, a->K, a->lda, a->ldw, 0, 0, 0, q};
    return launch_gemm<P3V_EPI_QKV>(p, s);
  }
  return P3V_OK;
}
