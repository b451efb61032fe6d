// Prompt-sized-but-short projections  C[M, N] = A[M, K] * W[N, K]^T  for 9 .. 256 rows (round 5): a 128-token chat prompt, the text
// group of a mixed batch, a 9..128-row decode batch.  At these M the GEMM is a WEIGHT STREAM (2 * 128 * N * K flops against N * K * 2
// bytes: ~13 GFLOP for the 100 MB of gate_up, both worth ~10-17 us), and what decides its speed is how many K-tiles a workgroup
// has on the wire: the 128 x 128-tile kernel (p3v_gemm.hip) holds ONE ahead and pays a DMA round trip (~1.3 us at this load) per
// K-tile -- gate_up 34 us + 5 us for the reduction of the split it needs to reach 256 workgroups, 2.9 TB/s.  Here:
//   tile      128(M) x 64(N) x 64(K): twice the workgroups per weight matrix, so gate_up (256 tiles) and qkv (144) run in one pass
//             and the others with a smaller split over K (o_proj 4, down 4 slices: one workgroup per CU, partials <= a quarter of
//             the weight bytes).  Every workgroup re-reads the A panel from L2: a launch stages W x 3 bytes, and the chip delivers
//             ~9.7 TB/s of staged bytes in this pattern whether the weights are hot in L2 or not (144 and 256 workgroups both settle
//             there) -- the kernel's bound; 128 x 128 tiles (W x 2 staged, half the workgroups) ran into the ~50 GB/s a single
//             CU's DMA path takes instead and were not faster on any shape (the template keeps the variant).
//   staging   LDS-DMA into a ring of 6 stages (A 16 KiB + W 8 KiB each), 5 in flight behind a counted vmcnt; one workgroup per CU
//             (<= 64 rows: 64-row tiles, A 8 KiB per stage -- a third less staged per K-tile, the same sums bit for bit: 60 -> 53 us
//             per decoder layer at 17 rows)
//   waves     8: four CONSUMERS (stacked along M, 32 rows x 64 columns each: acc[2][4]; W fragment first
//             in the MFMA, so the epilogue is the register-direct one of the other GEMM kernels, p3v_gemm256_epi.h) and four
//             PRODUCERS that only issue the DMA and wait for it; one barrier per K-tile hands a landed stage over and a drained
//             one back.  (With the MFMA waves issuing their own DMA a K-tile cost the SUM of its DMA issue, ~800 cycles for 24 KB,
//             and its fragment reads + MFMAs, ~400.)
//   K slices  gridDim.z; fp32 partials [S, M, ldp] in the caller's workspace, added in slice order by k_splitk_reduce (p3v_gemm.hip)
// (tools/scratch/gemm_skinny_regw_r5.hip: the variant that keeps each wave's W fragments out of LDS -- global loads straight into a
// register ring 8 K-tiles deep, A alone through LDS -- measured slower on every shape: profiles/r05_skinny_gemm.txt.)
#include "p3v_gemm_qkv.h"

#define SK_BM 128
#define SK_BK 64
#define SK_TN 64                    // tile width the launcher uses (TN = 128: measured, not faster -- see the header)
#ifndef SK_ABL
#define SK_ABL 0                    // timing experiments: 1 = A always K-tile 0 (one L2/L1-hot tile), 2 = W likewise, 3 = no MFMAs after K-tile 0
#endif
#define SK_A_BYTES (SK_BM * SK_BK * 2)

typedef __attribute__((address_space(3))) void* lptr_t;

template <int TN, int TM = SK_BM> struct SkCfg {
  static_assert(TM == 128 || (TM == 64 && TN == 64), "64-row tiles: prompts / batches of <= 64 rows, 64-column tiles");
  static constexpr int NS = TN == 64 ? 6 : 4;                  // ring stages (NS - 1 ahead)
  static constexpr int A_BYTES = TM * SK_BK * 2, W_BYTES = TN * SK_BK * 2, STAGE = A_BYTES + W_BYTES, LDS = NS * STAGE;
  static constexpr int AQ = TM / 32, WQ = TN / 32;             // A / W pieces (8 rows x 128 B) per producer wave and stage
  static constexpr int PER = AQ + WQ;                          // DMA instructions per producer wave and stage
  static constexpr int NI = TN == 64 ? TM / 64 : 4;            // 16-row blocks per consumer wave (its 64 columns: 4 blocks)
};

// s_waitcnt vmcnt(PER * n) for a uniform run-time n = stages that may stay in flight (PER = 6 or 8 DMA instructions per stage)
#define SK_WAIT_CASE(k, c) case k: asm volatile("s_waitcnt vmcnt(" #c ")" ::: "memory"); break;
template <int PER>
__device__ __forceinline__ void sk_wait_stages(int n) {
  if (PER == 6) {
    switch (n) {
      SK_WAIT_CASE(1, 6) SK_WAIT_CASE(2, 12) SK_WAIT_CASE(3, 18) SK_WAIT_CASE(4, 24)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (PER == 4) {
    switch (n) {
      SK_WAIT_CASE(1, 4) SK_WAIT_CASE(2, 8) SK_WAIT_CASE(3, 12) SK_WAIT_CASE(4, 16)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    static_assert(PER == 4 || PER == 6 || PER == 8, "");
    switch (n) {
      SK_WAIT_CASE(1, 8) SK_WAIT_CASE(2, 16) SK_WAIT_CASE(3, 24) SK_WAIT_CASE(4, 32)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

struct SkinnyP {
  const bf16_t* A; const bf16_t* W; void* out; const void* resid;
  int M, N, K, lda, ldw, ldo;      // N = output columns (SiLU: W holds 2N rows, gate rows then up rows)
  int kslice;                      // 0: one pass with the epilogue; else K columns per slice, fp32 partials
  QkvP q;                          // P3V_EPI_QKV only (p3v_gemm_qkv.h): head split + rotation + KV append in the epilogue
};

#define P3V_EPI_QKV 100            // internal, as in the other GEMM kernels

template <int EPI, bool PART, int TN, int TM = SK_BM>
__global__ void __launch_bounds__(512, 1) k_gemm_skinny(SkinnyP p) {
  typedef SkCfg<TN, TM> C;
  static_assert(TM == SK_BM || EPI != P3V_EPI_QKV, "the qkv epilogue's V tiles are 128 dimensions tall");
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  constexpr int NS = C::NS, NI = C::NI;
  static_assert(NS - 2 <= 4 && (NS - 2) * C::PER < 64, "sk_wait_stages covers 4 stages in flight");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool QKV = EPI == P3V_EPI_QKV;
  static_assert(!QKV || (TN == 64 && !PART), "the qkv epilogue: 64-column tiles, one pass");
  constexpr int n_out_tile = SILU ? TN / 2 : TN;
  int n0 = blockIdx.x * n_out_tile, m0 = blockIdx.y * TM;
  const int kz = blockIdx.z;
  const int nk = (PART ? p.kslice : p.K) / SK_BK, kt0 = kz * nk;
  // P3V_EPI_QKV (p3v_gemm_qkv.h), a one-dimensional grid: first the Q and K tiles (128 tokens x 32 rotation pairs, W rows fetched in
  // pair order), M-tile-major; then the V tiles with the operand roles swapped (128 V dimensions on the tile's A side, 64 tokens on
  // its B side), so that a lane ends up with consecutive TOKENS of one dimension for the transposed cache.
  bool swapped = false, is_k = false;
  int row0 = 0;
  if (QKV) {
    const int nq_t = p.q.nh * p.q.hd / 64, nk_t = p.q.nkv * p.q.hd / 64, nqk = nq_t + nk_t, mt = (p.M + SK_BM - 1) / SK_BM;
    int t = blockIdx.x;
    if (t < nqk * mt) {
      const int n_t = t % nqk;
      m0 = (t / nqk) * SK_BM;
      is_k = n_t >= nq_t;
      row0 = is_k ? p.q.nh * p.q.hd : 0;
      n0 = (n_t - (is_k ? nq_t : 0)) * 64;                     // first column INSIDE the region
    } else {
      t -= nqk * mt;
      const int nv_t = p.q.nkv * p.q.hd / SK_BM;
      swapped = true;
      row0 = (p.q.nh + p.q.nkv) * p.q.hd;
      n0 = (t % nv_t) * SK_BM;                                 // first V dimension of the tile (its A side)
      m0 = (t / nv_t) * 64;                                    // first token (its B side)
    }
  }

  if (wave >= 4) {
    // ---- PRODUCERS: wave pw stages 4 A pieces and WQ W pieces per K-tile, a piece = 8 tile rows x 128 B (1 KiB), the 16-byte
    // chunks of a row XOR-swizzled on the SOURCE side (the DMA destination is lane-linear).
    // SiLU: a 64-column group of the tile = gate rows (blocks 0, 1) and the up rows (blocks 2, 3) of the same 32 output columns.
    const int srow = lane >> 3, schunk = lane & 7, pw = wave - 4;
    unsigned a_src[4], w_src[4];                                 // (fixed sizes: see the note in stage())
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = rs_a;
    if (QKV && swapped) { rs_a = rs_w; rs_w = rs_x; }           // the tile's A side reads W rows, its B side token rows
#pragma unroll
    for (int q = 0; q < C::AQ; ++q) {
      const int r = (pw * C::AQ + q) * 8 + srow;
      if (QKV && swapped) {
        a_src[q] = (unsigned)(((size_t)(row0 + n0 + r) * p.ldw + ((schunk ^ (r & 7)) * 8)) * 2);
      } else {
        const int ar = min(m0 + r, p.M - 1);
        a_src[q] = (unsigned)(((size_t)ar * p.lda + ((schunk ^ (r & 7)) * 8)) * 2);
      }
    }
#pragma unroll
    for (int q = 0; q < C::WQ; ++q) {
      const int r = (pw * C::WQ + q) * 8 + srow;
      if (QKV && swapped) {
        const int ar = min(m0 + r, p.M - 1);
        w_src[q] = (unsigned)(((size_t)ar * p.lda + ((schunk ^ (r & 7)) * 8)) * 2);
        continue;
      }
      int br;
      if (SILU) br = min(n0 + (r >> 6) * 32 + ((r >> 4) & 1) * 16 + (r & 15), p.N - 1) + ((r >> 5) & 1) * p.N;
      else if (QKV) br = qkv_pair_row(p.q, row0, n0 >> 1, r);
      else br = min(n0 + r, p.N - 1);
      w_src[q] = (unsigned)(((size_t)br * p.ldw + ((schunk ^ (r & 7)) * 8)) * 2);
    }
    auto stage = [&](int kt, int slot) {
      // (every argument of the builtin is a local of NON-DEPENDENT type: a type-dependent one defers the check to instantiation,
      // where the HOST pass does not know the builtin and hipcc silently drops the kernel's host stub)
      unsigned char* base = smem + slot * (int)C::STAGE;
      const int koff = (kt0 + kt) * (SK_BK * 2), koff_a = SK_ABL == 1 ? 0 : koff, koff_w = SK_ABL == 2 ? 0 : koff;
      const int wq = C::WQ, aq = C::AQ, a_bytes = C::A_BYTES;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < aq) {
          lptr_t dst = (lptr_t)(base + (pw * aq + q) * 1024);
          const unsigned vo = a_src[q];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, vo, koff_a, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < wq) {
          lptr_t dst = (lptr_t)(base + a_bytes + (pw * wq + q) * 1024);
          const unsigned vo = SK_ABL == 2 ? a_src[q] : w_src[q];
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, vo, koff_w, 0, 0);
        }
      }
    };
    for (int st = 0; st < min(NS - 1, nk); ++st) stage(st, st);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      sk_wait_stages<C::PER>(min(NS - 2, nk - 1 - kt));        // stage kt has landed when only the younger stages are still out
      __builtin_amdgcn_s_barrier();                             // tells the consumers; and they are done reading kt - 1
      if (kt + NS - 1 < nk) stage(kt + NS - 1, slot == 0 ? NS - 1 : slot - 1);   // ... whose slot takes K-tile kt + NS - 1
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
    return;
  }

  // ---- CONSUMERS: wave (wr, wc) owns NI x 16 rows from wr * NI * 16 and the 64 columns of group wc
  const int wr = TN == 64 ? wave : wave >> 1, wc = TN == 64 ? 0 : wave & 1;
  f32x4_t acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  {
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();                             // K-tile kt is in LDS (every producer waited for its pieces)
      const unsigned char* ta = smem + slot * C::STAGE;
      const unsigned char* tb = ta + C::A_BYTES;
      bf16x8_t af[2][NI], bfr[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = (wr * NI + i) * 16 + frow;
          af[kk][i] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = wc * 64 + j * 16 + frow;
          bfr[kk][j] = *(const bf16x8_t*)(tb + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);                        // all fragment reads out before the first MFMA (counted lgkmcnt)
#pragma unroll
      for (int kk = 0; kk < (SK_ABL == 3 ? (kt == 0 ? 2 : 0) : 2); ++kk)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);   // W first: transposed block
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
  }

  // ---- epilogue, straight from the accumulators: lane (fc = lane & 15, fq = lane >> 4) holds columns 4 fq .. 4 fq + 3 of block
  //      row fc (see p3v_gemm256_epi.h)
  if constexpr (QKV) {
    if (!swapped) qkv_epilogue_rot<NI>(p.q, acc, nullptr, is_k, row0, n0 >> 1, m0 + wr * NI * 16, p.M, lane);
    else qkv_epilogue_vt<NI>(p.q, acc, nullptr, n0 + wr * NI * 16, m0, p.M, lane);
    return;
  }
  const int fc = lane & 15, fq = lane >> 4;
  const int mrow0 = m0 + wr * NI * 16 + fc;
  if constexpr (PART) {
    // fp32 partial of slice kz, [M, ldp] with ldp = W rows: SiLU keeps [gate | up] as 2N plain columns (k_splitk_reduce's layout)
    const int ldp = SILU ? 2 * p.N : p.N;
    float* part = (float*)p.out + (size_t)kz * p.M * ldp;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = mrow0 + i * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nn = SILU ? n0 + wc * 32 + (j & 1) * 16 + fq * 4 : n0 + wc * 64 + j * 16 + fq * 4;
        if (m < p.M && nn < p.N)
          *(float4*)(part + (size_t)m * ldp + nn + (SILU ? (j >> 1) * p.N : 0)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
  if constexpr (SILU) {
    const int n = n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
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
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = mrow0 + i * 16;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      uint32_t pk[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * jp + jj;
        pk[jj][0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]), pk[jj][1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
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

// K slices for a shape on this kernel: 0 = not one of its shapes; 1 = one pass; S > 1 = S slices + the reduction launch.
// One workgroup per CU: the largest S <= 8 whose tiles * S still fit one round of the chip (slices of >= 6 K-tiles).
extern "C" int p3v_gemm_rows_slices(int M, int N, int K, int epilogue);            // p3v_gemm_rows.hip: 9 .. 32 rows, weights straight to registers
int p3v_gemm_rows_launch(const p3v_gemm_args_t* a, float* part, hipStream_t s);
static int sk_slices(int M, int N, int K, int epilogue);
int p3v_gemm_skinny_slices(int M, int N, int K, int epilogue) {
  const int rs = p3v_gemm_rows_slices(M, N, K, epilogue);
  return rs ? rs : sk_slices(M, N, K, epilogue);
}
static int sk_slices(int M, int N, int K, int epilogue) {
  const P3vTuning& t = p3v_tuning();
  const bool silu = epilogue == P3V_EPI_SILU_MUL;
  if (t.gemm_no_skinny || M <= 8 || M > t.gemm_skinny_max_m || K % SK_BK || N % (silu ? 32 : 64)) return 0;
  if (epilogue != P3V_EPI_NONE && epilogue != P3V_EPI_RESID_BF16 && !silu) return 0;
  const int tiles = p3v_cdiv(M, SK_BM) * ((silu ? 2 * N : N) / SK_TN);
  if (t.gemm_skinny_s > 0) return K % (t.gemm_skinny_s * SK_BK) ? 1 : t.gemm_skinny_s;
  int best = 1;
  for (int S = 2; S <= 8; ++S)
    if (K % (S * SK_BK) == 0 && K / S >= 6 * SK_BK && tiles * S <= t.gemm_splitk_wgs) best = S;
  return best;
}

template <int EPI, int TM>
static int launch_skinny_tm(const SkinnyP& p, int S, hipStream_t s) {
  typedef SkCfg<SK_TN, TM> C;
  static bool attr_set[2] = {false, false};
  const bool part = S > 1;
  if (!attr_set[part]) {
    const void* fn = part ? (const void*)k_gemm_skinny<EPI, true, SK_TN, TM> : (const void*)k_gemm_skinny<EPI, false, SK_TN, TM>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return P3V_ERR_HIP;
    attr_set[part] = true;
  }
  const int n_tile = EPI == P3V_EPI_SILU_MUL ? SK_TN / 2 : SK_TN;
  const dim3 grid(p.N / n_tile, p3v_cdiv(p.M, TM), S);
  if (part) hipLaunchKernelGGL((k_gemm_skinny<EPI, true, SK_TN, TM>), grid, dim3(512), C::LDS, s, p);
  else hipLaunchKernelGGL((k_gemm_skinny<EPI, false, SK_TN, TM>), grid, dim3(512), C::LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
// <= 64 rows: 64-row tiles (the A panel is two thirds of what a 128-row tile stages per K-tile; half of it would be clamped copies)
template <int EPI>
static int launch_skinny(const SkinnyP& p, int S, hipStream_t s) {
  return p.M <= 64 && !p3v_tuning().gemm_skinny_tm128 ? launch_skinny_tm<EPI, 64>(p, S, s) : launch_skinny_tm<EPI, SK_BM>(p, S, s);
}

int p3v_splitk_reduce(const float* part, const p3v_gemm_args_t* a, int S, hipStream_t s);   // p3v_gemm.hip

// the K-slice launch alone (fp32 partials into a->ws; *S_out slices): P3V_ERR_UNSUPPORTED, nothing launched, unless the shape is
// one this kernel splits and the workspace holds the partials
int p3v_gemm_skinny_partials(const p3v_gemm_args_t* a, int* S_out, hipStream_t s) {
  const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
  const int w_rows = silu ? 2 * a->N : a->N;
  const int RS = a->bias ? 0 : p3v_gemm_rows_slices(a->M, a->N, a->K, a->epilogue);
  if (RS > 1 && a->ws && a->ws_bytes >= (int64_t)RS * a->M * w_rows * 4) {        // 9 .. 32 rows: the register-streaming kernel's K slices
    if ((uintptr_t)a->ws & 15) return P3V_ERR_ARG;
    *S_out = RS;
    return p3v_gemm_rows_launch(a, (float*)a->ws, s);
  }
  if (RS == 1) return P3V_ERR_UNSUPPORTED;                       // (runs in one pass: no partials)
  const int S = sk_slices(a->M, a->N, a->K, a->epilogue);
  if (S <= 1 || !a->ws || a->ws_bytes < (int64_t)S * a->M * w_rows * 4) return P3V_ERR_UNSUPPORTED;
  if ((uintptr_t)a->ws & 15) return P3V_ERR_ARG;
  *S_out = S;
  const SkinnyP p = {a->A, a->W, a->ws, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, a->K / S, {}};
  if (silu) return launch_skinny<P3V_EPI_SILU_MUL>(p, S, s);
  if (a->epilogue == P3V_EPI_RESID_BF16) return launch_skinny<P3V_EPI_RESID_BF16>(p, S, s);
  return launch_skinny<P3V_EPI_NONE>(p, S, s);
}

// P3V_ERR_UNSUPPORTED: not a shape for this kernel (the caller goes on to the other GEMM paths).  A split that finds no workspace of
// p3v_gemm_ws_bytes() runs as one pass.
int p3v_gemm_skinny_try(const p3v_gemm_args_t* a, hipStream_t s) {
  if (p3v_gemm_skinny_slices(a->M, a->N, a->K, a->epilogue) == 0) return P3V_ERR_UNSUPPORTED;
  if (p3v_gemm_rows_slices(a->M, a->N, a->K, a->epilogue) == 1) {                  // 9 .. 32 rows, one pass
    const int rr = p3v_gemm_rows_launch(a, nullptr, s);
    if (rr != P3V_ERR_UNSUPPORTED) return rr;
  }
  int S = 1;
  const int rc = p3v_gemm_skinny_partials(a, &S, s);
  if (rc == P3V_OK) return p3v_splitk_reduce((const float*)a->ws, a, S, s);
  if (rc != P3V_ERR_UNSUPPORTED) return rc;
  if (sk_slices(a->M, a->N, a->K, a->epilogue) == 0) return P3V_ERR_UNSUPPORTED;   // (a rows shape whose slices found no workspace and that the tile kernel does not take)
  const SkinnyP p = {a->A, a->W, a->out, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, 0, {}};
  if (a->epilogue == P3V_EPI_SILU_MUL) return launch_skinny<P3V_EPI_SILU_MUL>(p, 1, s);
  if (a->epilogue == P3V_EPI_RESID_BF16) return launch_skinny<P3V_EPI_RESID_BF16>(p, 1, s);
  return launch_skinny<P3V_EPI_NONE>(p, 1, s);
}

// ---- the qkv projection of a short prompt with the head split, the rotation and the KV append in the epilogue (p3v_gemm_qkv's
// 17 .. 256-row case): one pass (the epilogue needs whole sums), Q / K tiles + V tiles in a one-dimensional grid.
// P3V_ERR_UNSUPPORTED (nothing launched) where the shape is not this kernel's.
int p3v_gemm_skinny_qkv(const p3v_gemm_args_t* a, const QkvP& q, hipStream_t s) {
  const P3vTuning& t = p3v_tuning();
  if (t.gemm_no_skinny || a->M <= 8 || a->M > t.gemm_skinny_max_m || a->K % SK_BK || a->bias) return P3V_ERR_UNSUPPORTED;
  if ((q.nh * q.hd) % 64 || (q.nkv * q.hd) % SK_BM) return P3V_ERR_UNSUPPORTED;   // whole Q / K tiles of 32 pairs, whole V tiles of 128 dims
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_gemm_skinny<P3V_EPI_QKV, false, SK_TN>, hipFuncAttributeMaxDynamicSharedMemorySize, SkCfg<SK_TN>::LDS) != hipSuccess)
      return P3V_ERR_HIP;
    attr_set = true;
  }
  const SkinnyP p = {a->A, a->W, nullptr, nullptr, a->M, a->N, a->K, a->lda, a->ldw, 0, 0, q};
  const int mt = p3v_cdiv(a->M, SK_BM), tiles = ((q.nh + q.nkv) * q.hd / 64) * mt + (q.nkv * q.hd / SK_BM) * p3v_cdiv(a->M, 64);
  hipLaunchKernelGGL((k_gemm_skinny<P3V_EPI_QKV, false, SK_TN>), dim3(tiles), dim3(512), SkCfg<SK_TN>::LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}
