// Projections  C[M, N] = x[M, K] * W[N, K]^T  for 9 .. 32 rows of x (round 6): a batched decode step of 9 .. 32 sequences -- the
// reference's own batched benchmark runs 15 (phi_3_vision_mlx.py:1226-1243) -- and any continuous-batching server.  At these M the
// projection is a WEIGHT STREAM; the 64-row tiles of p3v_gemm_skinny.hip stage every weight byte through LDS next to a re-read x
// panel (W x 3 bytes staged per launch, 2.1 - 4.4 TB/s of weights, B = 9 slower per token than B = 8).  This kernel is k_gemv_mfma8
// (p3v_gemv.hip: 5.7 TB/s on gate_up at 8 rows) carried to 16 / 32 rows:
//   * weights go HBM -> registers and nowhere else, eight FULL 128-byte lines per load instruction: the 16 MFMA rows of a fragment are
//     8 weight rows x 2 k-halves (A row i = 2 r + h: lane (i, g) holds the 16-byte chunk 2 g + h of a 64-element line of row r), so
//     lane pairs read 32 contiguous bytes and an instruction covers 8 whole lines (tools/scratch/frag_stream.hip: 5.5 TB/s, against 4.0
//     for the half-line pattern of a plain 16-row fragment);
//   * x lives in REGISTERS as ready MFMA B fragments -- wave w keeps its K quarter of all rows for the whole launch, 96 VGPRs per 16
//     rows at K = 3072 -- in the two k-orders the interleaved A rows need: B_h'[k-slot g][column m] = x[m][chunk 2 g + h'].  Per weight
//     line the wave issues two MFMAs per 16 rows of x (h' = 0, 1); C row 2 r + h of the h' = h product is the dot-product piece of
//     weight row r over that k-half, the other half of each product is discarded (the matrix cores are idle anyway: 2 x 2 x 12 MFMAs
//     per 12 KB of weights at 32 rows), and the two halves are added IN the lane (no DPP, no LDS);
//   * a workgroup walks `sets` of 16 output columns (SiLU: 8 gate + the 8 matching up rows), the whole K slice of the NEXT set being
//     requested stage by stage as the current one is consumed; the K quarters of a set meet in a double-buffered LDS block (one
//     barrier per set), where the epilogue -- none / residual / SiLU * up, or fp32 K-slice partials for p3v_splitk_reduce /
//     k_splitk_reduce_norm -- is applied with k_gemv_mfma8's arithmetic;
//   * no LDS for operands, so nothing limits M x K but registers: K = 3072 in one pass up to 32 rows (qkv, gate_up); o_proj and down
//     (192 sets of 16 columns each) run as 4 K slices x 64 workgroups so that all 256 CUs stream (their reduction launch exists
//     anyway: it also writes the next RMSNorm).
// The RMSNorm of the input is the caller's (at these M the model runs it in the previous projection's reduction launch).
#include "p3v_common.h"

struct RowsP {
  const bf16_t* x; const bf16_t* W; void* out; const bf16_t* resid; float* part;
  int M, N, K, lda, ldw, ldo, epi, n_sets, w_rows;
};

typedef std::integral_constant<int, 0> RIC0;
typedef std::integral_constant<int, 1> RIC1;
typedef std::integral_constant<int, 2> RIC2;
typedef std::integral_constant<int, 3> RIC3;

template <bool SILU, int NW, int NST, int MT, bool PART>
__global__ void __launch_bounds__(NW * 64, 1) k_gemm_rows(RowsP p) {
  constexpr int KQ = NST * 256, KWG = KQ * NW, NL = NST * 4;    // a wave's K quarter, the workgroup's K slice, weight lines per quarter
  constexpr int MB = 16 * MT;                                   // x rows held
  __shared__ float cpart[2][NW * 2 * 8 * MB];                   // [parity][wave][row set][weight row][x row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int r8 = li >> 1, chunk = 2 * g + (lane & 1);            // weight row in the set; 16-byte chunk of a 64-element line
  const int z = blockIdx.y;                                      // K slice
  const int k_lo = z * KWG + wave * KQ;

  auto row_ptrs = [&](int set, const bf16_t*& w0, const bf16_t*& w1) {
    const int n_base = set * (SILU ? 8 : 16);
    const int row0 = min(n_base + r8, p.N - 1);
    const int row1 = SILU ? p.N + row0 : min(n_base + 8 + r8, p.N - 1);
    w0 = p.W + (size_t)row0 * p.ldw + k_lo + 8 * chunk;
    w1 = p.W + (size_t)row1 * p.ldw + k_lo + 8 * chunk;
  };
  int set = blockIdx.x;
  const bf16_t *w0, *w1;
  row_ptrs(set, w0, w1);

  u32x4_t wa[NST][8];                                            // stage = 4 lines x 2 row sets; one whole quarter of a set pair
  auto issue = [&](const bf16_t* a0, const bf16_t* a1, auto stc) {
    constexpr int st = decltype(stc)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wa[st][2 * q] = __builtin_nontemporal_load((const u32x4_t*)(a0 + (st * 4 + q) * 64));
      wa[st][2 * q + 1] = __builtin_nontemporal_load((const u32x4_t*)(a1 + (st * 4 + q) * 64));
    }
  };

  // ---- the first weight stage goes out first (its HBM round trip covers the x loads, which come from L2), then this wave's quarter of
  // x as B fragments: column li of block mt is x row 16 mt + li, k-slot g of line q holds chunk 2 g + h'
  issue(w0, w1, RIC0{});
  u32x4_t xf[MT][NL][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const bf16_t* xr = p.x + (size_t)min(16 * mt + li, p.M - 1) * p.lda + k_lo + 16 * g;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      xf[mt][q][0] = *(const u32x4_t*)(xr + q * 64);
      xf[mt][q][1] = *(const u32x4_t*)(xr + q * 64 + 8);
    }
  }
  if constexpr (NST > 1) issue(w0, w1, RIC1{});
  if constexpr (NST > 2) issue(w0, w1, RIC2{});
  if constexpr (NST > 3) issue(w0, w1, RIC3{});
  static_assert(NST <= 4, "unrolled for at most four stages");

  int par = 0;
  // two straight-line copies of the set body (as k_gemv_mfma8): with a further set to request, and the last one -- a branch around the
  // refills makes the compiler's counted vmcnt ignore them, and the last stage of every set then waits for loads issued a moment before
  auto do_set = [&](auto refillc) {
    constexpr bool REFILL = decltype(refillc)::value;
    const int nset = set + (int)gridDim.x;
    const bf16_t *w0n = nullptr, *w1n = nullptr;
    if constexpr (REFILL) row_ptrs(nset, w0n, w1n);
    f32x4_t acc[2][MT][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[s][mt][0] = acc[s][mt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st < NST) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, wa[st][2 * q + s]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              acc[s][mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, xf[mt][st * 4 + q][0]), acc[s][mt][0], 0, 0, 0);
              acc[s][mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, xf[mt][st * 4 + q][1]), acc[s][mt][1], 0, 0, 0);
            }
          }
        }
        if constexpr (REFILL) { issue(w0n, w1n, stc); __builtin_amdgcn_sched_barrier(0); }   // (pinned behind this stage's MFMAs)
      }
    };
    step(RIC0{}); step(RIC1{}); step(RIC2{}); step(RIC3{});

    // ---- C[4 g + e][li]: rows 4 g + {0, 2} of the h' = 0 product and 4 g + {1, 3} of the h' = 1 product are the two k-halves of weight
    // rows 2 g and 2 g + 1 against x row li of the block: added in the lane, then the NW K quarters through LDS
    float* cp = cpart[par];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float e0 = acc[s][mt][0][0] + acc[s][mt][1][1], e1 = acc[s][mt][0][2] + acc[s][mt][1][3];
        cp[((wave * 2 + s) * 8 + 2 * g) * MB + 16 * mt + li] = e0;
        cp[((wave * 2 + s) * 8 + 2 * g + 1) * MB + 16 * mt + li] = e1;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // (the weight loads of the set ahead stay in flight across it)
    constexpr int ITEMS = (SILU ? 8 : 16) * MB;                  // (output column, x row) pairs of a set
#pragma unroll 1
    for (int idx = tid; idx < ITEMS; idx += NW * 64) {
      const int R = idx & 7, sub = SILU ? 0 : (idx >> 3) & 1, m = SILU ? idx >> 3 : idx >> 4;
      const int n = set * (SILU ? 8 : 16) + sub * 8 + R;
      if (m < p.M && n < p.N) {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          v0 += cp[((w * 2 + sub) * 8 + R) * MB + m];
          if (SILU) v1 += cp[((w * 2 + 1) * 8 + R) * MB + m];
        }
        if (PART) {                                              // fp32 partials [slice, M, W rows] (SiLU: gate columns, then up columns)
          float* dst = p.part + ((size_t)z * p.M + m) * p.w_rows;
          dst[n] = v0;
          if (SILU) dst[p.N + n] = v1;
        } else if (SILU) {
          const float gt = bf16_round(v0), up = bf16_round(v1);
          ((bf16_t*)p.out)[(size_t)m * p.ldo + n] = f32_to_bf16(bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up);
        } else if (p.epi == P3V_EPI_RESID_BF16) {
          ((bf16_t*)p.out)[(size_t)m * p.ldo + n] = f32_to_bf16(bf16_to_f32(p.resid[(size_t)m * p.ldo + n]) + bf16_round(v0));
        } else {
          ((bf16_t*)p.out)[(size_t)m * p.ldo + n] = f32_to_bf16(v0);
        }
      }
    }
    set = nset;
    par ^= 1;
  };
  while (set + (int)gridDim.x < p.n_sets) do_set(std::true_type{});
  do_set(std::false_type{});
}

// ---- shapes: K slices S and the (waves, stages) of a workgroup's slice.  0 = not this kernel's.
struct RowsPlan { int S, NW, NST; };
static RowsPlan rows_plan(int M, int N, int K, int epilogue) {
  const P3vTuning& t = p3v_tuning();
  const bool silu = epilogue == P3V_EPI_SILU_MUL;
  RowsPlan none = {0, 0, 0};
  if (!t.gemm_rows || M <= 8 || M > 32 || N % (silu ? 8 : 16)) return none;
  if (epilogue != P3V_EPI_NONE && epilogue != P3V_EPI_RESID_BF16 && !silu) return none;
  const int n_sets = N / (silu ? 8 : 16);
  if (K == 3072) return n_sets >= 512 ? RowsPlan{1, 4, 3} : RowsPlan{4, 3, 1};     // qkv / gate_up in one pass; o_proj: 4 slices of 768
  if (K == 8192) return RowsPlan{4, 4, 2};                                          // down: 4 slices of 2048
  return none;
}

extern "C" int p3v_gemm_rows_slices(int M, int N, int K, int epilogue) { return rows_plan(M, N, K, epilogue).S; }

template <bool SILU, int NW, int NST, int MT, bool PART>
static int launch_rows(const RowsP& p, int S, hipStream_t s) {
  // workgroups along N: all of a one-pass launch's CUs, a quarter of them per K slice otherwise (a workgroup's x quarter is loaded once
  // and amortised over its sets: fewer, longer workgroups keep the x traffic from L2 below the weight bytes)
  const int max_wg = p3v_cdiv(256, S);
  const int per = p3v_cdiv(p.n_sets, max_wg), gx = p3v_cdiv(p.n_sets, per);
  hipLaunchKernelGGL((k_gemm_rows<SILU, NW, NST, MT, PART>), dim3(gx, S), dim3(NW * 64), 0, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

template <bool SILU, int NW, int NST, bool PART>
static int launch_rows_mt(const RowsP& p, int S, hipStream_t s) {
  return p.M <= 16 ? launch_rows<SILU, NW, NST, 1, PART>(p, S, s) : launch_rows<SILU, NW, NST, 2, PART>(p, S, s);
}

// one pass with the epilogue (S == 1), or the K-slice partials alone (part != nullptr): P3V_ERR_UNSUPPORTED when the shape is not planned
int p3v_gemm_rows_launch(const p3v_gemm_args_t* a, float* part, hipStream_t s) {
  const RowsPlan pl = rows_plan(a->M, a->N, a->K, a->epilogue);
  if (pl.S == 0 || a->bias || (pl.S > 1) != (part != nullptr)) return P3V_ERR_UNSUPPORTED;
  const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
  const RowsP p = {a->A, a->W, a->out, (const bf16_t*)a->resid, part, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, a->epilogue,
                   a->N / (silu ? 8 : 16), silu ? 2 * a->N : a->N};
  if (pl.S == 1) return silu ? launch_rows_mt<true, 4, 3, false>(p, 1, s) : launch_rows_mt<false, 4, 3, false>(p, 1, s);
  if (silu) return P3V_ERR_UNSUPPORTED;                          // (no planned shape splits a SiLU projection)
  if (pl.NW == 3) return launch_rows_mt<false, 3, 1, true>(p, pl.S, s);
  return launch_rows_mt<false, 4, 2, true>(p, pl.S, s);
}
