// Prompt-sized-but-short projections  C[M, N] = A[M, K] * W[N, K]^T  for 17 .. 256 rows (round 5): a 128-token chat prompt, the text
// group of a mixed batch, a 17..128-row decode batch.  At these M the GEMM is a WEIGHT STREAM (2 * 128 * N * K flops against N * K * 2
// bytes: ~13 GFLOP for the 100 MB of gate_up, both worth ~10-17 us), and what decides its speed is how many weight bytes the chip
// keeps in flight (8 TB/s x ~2 us of loaded latency = 12-16 MB), not the MFMA schedule.  The 128 x 128-tile kernel (p3v_gemm.hip)
// holds ONE K-tile ahead per workgroup and needs a split over K to reach 256 workgroups even for gate_up: 34 + 5 us, 2.9 TB/s.  A first
// version of this file staged W through a 3-stage LDS ring beside A (8 KB of W per stage, two ahead: 4 MB on the wire): 31 us.  Now:
//   tile      128(M) x 64(N) x 64(K); 4 waves, each owning 16 of the 64 columns for all 128 rows (acc[8]): a wave's W fragments are
//             ITS OWN, so they never touch LDS -- non-temporal global loads straight into the MFMA operand registers, a ring of
//             SK_D K-tiles (1 KB each per wave-load) deep: 48 KB of weights on the wire per workgroup, 12-24 MB over the chip
//   A         (786 KB for 128 x 3072: L2-resident, every workgroup re-reads it) through an LDS-DMA ring of SK_NSA stages of 16 KiB,
//             XOR-swizzled like the other GEMM kernels; one barrier per K-tile
//   waits     one counted vmcnt per K-tile covers both rings (loads retire in order; the count is the number of younger DMA / W loads)
//   K slices  gridDim.z; fp32 partials [S, M, ldp] in the caller's workspace, added in slice order by k_splitk_reduce (p3v_gemm.hip):
//             gate_up (256 tiles) needs none, qkv 2, o_proj 6, down 8 -- partials <= a quarter of the weight bytes
//   SiLU      a wave's 16 MFMA columns are 8 gate rows and the 8 up rows of the same outputs; v_permlane32_swap brings a column's two
//             halves into one lane
// LDS fragment reads (16 KB per wave and K-tile for 16 MFMAs) bound the K loop at ~0.21 us per K-tile and CU: 10 us for gate_up, 6 for
// qkv -- under the weight stream's time for every decoder shape.
#include "p3v_gemm_qkv.h"

#define SK_BM 128
#define SK_BN 64
#define SK_BK 64
#ifndef SK_NSA
#define SK_NSA 4                    // A stages in LDS (3 ahead)
#endif
#ifndef SK_D
#define SK_D 8                      // K-tiles of W in registers per wave (the K slice is a multiple of it)
#endif
#define SK_A_BYTES (SK_BM * SK_BK * 2)
#define SK_LDS (SK_NSA * SK_A_BYTES)

#include <type_traits>
#include <utility>
template <int N, class F, int... I>
__device__ __forceinline__ void p3v_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void p3v_static_for(F&& f) { p3v_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int N>
__device__ __forceinline__ void p3v_wait_vmcnt_c() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// vm operations issued after the A pieces of K-tile kt in a K loop of nk tiles: iteration j issues, after its barrier, the 4 A pieces
// of K-tile j + NSA - 1 and then, after its MFMAs, the 2 W loads of K-tile j + D (each only while that tile exists); the prologue
// issues the whole W ring, then A stages 0 .. NSA - 2.
constexpr int sk_group(int j, int nk, int NSA, int D) { return (j + NSA - 1 < nk ? 4 : 0) + (j + D < nk ? 2 : 0); }
constexpr int sk_younger(int kt, int nk, int NSA, int D) {
  int y = 0;
  if (kt <= NSA - 2) {
    y = 4 * ((NSA - 1 < nk ? NSA - 1 : nk) - 1 - kt);
    for (int j = 0; j < kt; ++j) y += sk_group(j, nk, NSA, D);
  } else {
    y = kt - NSA + 1 + D < nk ? 2 : 0;
    for (int j = kt - NSA + 2; j < kt; ++j) y += sk_group(j, nk, NSA, D);
  }
  return y;
}

typedef __attribute__((address_space(3))) void* lptr_t;

struct SkinnyP {
  const bf16_t* A; const bf16_t* W; void* out; const void* resid;
  int M, N, K, lda, ldw, ldo;      // N = output columns (SiLU: W holds 2N rows, gate rows then up rows)
  int kslice;                      // 0: one pass with the epilogue; else K columns per slice, fp32 partials
};

template <int EPI, bool PART>
__global__ void __launch_bounds__(256, 2) k_gemm_skinny(SkinnyP p) {
  constexpr bool SILU = EPI == P3V_EPI_SILU_MUL;
  constexpr int NSA = SK_NSA, D = SK_D;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int n_out_tile = SILU ? SK_BN / 2 : SK_BN;
  const int n0 = blockIdx.x * n_out_tile, m0 = blockIdx.y * SK_BM, kz = blockIdx.z;
  const int nk = (PART ? p.kslice : p.K) / SK_BK, kt0 = kz * nk;
  const int frow = lane & 15, fchunk = lane >> 4;

  // ---- A: wave w issues 4 DMA pieces per stage, piece q covers tile rows (w*4+q)*8 .. +8 (1 KiB)
  const int srow = lane >> 3, schunk = lane & 7;
  unsigned a_src[4];
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffff, 0x00020000);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (wave * 4 + q) * 8 + srow;
    const int ar = min(m0 + r, p.M - 1);
    a_src[q] = (unsigned)(((size_t)ar * p.lda + ((schunk ^ (r & 7)) * 8)) * 2);
  }
  auto stage_a = [&](int kt, int slot) {
    unsigned char* base = smem + slot * SK_A_BYTES;
    const int koff = (kt0 + kt) * (SK_BK * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(base + (wave * 4 + q) * 1024), 16, a_src[q], koff, 0, 0);
  };
  // ---- W: the lane's MFMA operand row.  SiLU: columns 0-7 of the wave are gate rows, 8-15 the up rows of the same 8 outputs
  const int wrow = SILU ? n0 + wave * 8 + (frow & 7) + (frow >> 3) * p.N : n0 + wave * 16 + frow;
  const bf16_t* wp = p.W + (size_t)wrow * p.ldw + (size_t)kt0 * SK_BK + fchunk * 8;
  u32x4_t wr[D][2];
  auto load_w = [&](int kt, auto dc) {
    constexpr int d = decltype(dc)::value;
#ifdef SK_NT
    wr[d][0] = __builtin_nontemporal_load((const u32x4_t*)(wp + kt * SK_BK));
    wr[d][1] = __builtin_nontemporal_load((const u32x4_t*)(wp + kt * SK_BK + 32));
#else
    wr[d][0] = *(const u32x4_t*)(wp + kt * SK_BK);             // (plain loads: the two halves of a 128-byte line come in two
    wr[d][1] = *(const u32x4_t*)(wp + kt * SK_BK + 32);        //  instructions, and a streaming hint makes the second one miss again)
#endif
  };

  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the W ring first (older than every A piece: one counted wait on the A stage covers the K-tile's W too)
  p3v_static_for<D>([&](auto dc) { load_w(decltype(dc)::value, dc); });
  p3v_static_for<NSA - 1>([&](auto sc) { stage_a(decltype(sc)::value, decltype(sc)::value); });
  // The K loop runs in blocks of D K-tiles (nk is a multiple of D: launcher), every block straight-line code: with a branch per
  // K-tile the compiler's own wait-count pass loses track of the register ring and puts vmcnt(0) in front of the MFMAs.
  //   MAIN blocks reload their ring slot with K-tile kt + D; the FINAL block (the last D K-tiles) does not, and stops staging A
  //   NSA - 1 tiles before the end.  `younger` = vm operations issued after the A pieces of K-tile kt (they may stay in flight),
  //   evaluated at compile time on a model K loop of 3 D tiles (first / middle / last block; 1 block when nk == D).
  int slot = 0;
  auto k_tile = [&](int kt, auto dc, auto ktm_c, auto nkm_c) {
    constexpr int d = decltype(dc)::value, ktm = decltype(ktm_c)::value, nkm = decltype(nkm_c)::value;
    constexpr int younger = sk_younger(ktm, nkm, NSA, D);
    p3v_wait_vmcnt_c<younger>();
    __builtin_amdgcn_s_barrier();                               // everybody's pieces of kt are in; everybody is done reading kt - 1
    if constexpr (ktm + NSA - 1 < nkm) stage_a(kt + NSA - 1, slot == 0 ? NSA - 1 : slot - 1);   // ... whose slot takes K-tile kt + NSA - 1
    const unsigned char* ta = smem + slot * SK_A_BYTES;
    bf16x8_t af[2][8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + frow;
        af[kk][i] = *(const bf16x8_t*)(ta + r * 128 + (((kk * 4 + fchunk) ^ (r & 7)) << 4));
      }
    __builtin_amdgcn_sched_barrier(0);                          // all 16 fragment reads out before the first MFMA (counted lgkmcnt)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wr[d][kk]), af[kk][i], acc[i], 0, 0, 0);   // W first
    __builtin_amdgcn_sched_barrier(0);                          // the ring slot is reloaded AFTER its MFMAs and after this tile's A pieces
    if constexpr (ktm + D < nkm) load_w(kt + D, dc);
    slot = slot == NSA - 1 ? 0 : slot + 1;
  };
  typedef std::integral_constant<int, 3 * D> NK3;
  typedef std::integral_constant<int, D> NK1;
  if (nk == D) {
    p3v_static_for<D>([&](auto dc) { k_tile(decltype(dc)::value, dc, dc, NK1{}); });
  } else {
    p3v_static_for<D>([&](auto dc) { k_tile(decltype(dc)::value, dc, dc, NK3{}); });
    for (int base = D; base + D < nk; base += D)
      p3v_static_for<D>([&](auto dc) { k_tile(base + decltype(dc)::value, dc, std::integral_constant<int, D + decltype(dc)::value>{}, NK3{}); });
    p3v_static_for<D>([&](auto dc) { k_tile(nk - D + decltype(dc)::value, dc, std::integral_constant<int, 2 * D + decltype(dc)::value>{}, NK3{}); });
  }

  // ---- epilogue, straight from the accumulators: lane (fc = lane & 15, fq = lane >> 4) holds wave columns 4 fq .. 4 fq + 3 of
  //      block row fc
  const int fc = lane & 15, fq = lane >> 4;
  if constexpr (PART) {
    // fp32 partial of slice kz, [M, ldp] with ldp = W rows: SiLU keeps [gate | up] as 2N plain columns (k_splitk_reduce's layout)
    const int ldp = SILU ? 2 * p.N : p.N;
    float* part = (float*)p.out + (size_t)kz * p.M * ldp;
    const int col = SILU ? n0 + wave * 8 + (fq & 1) * 4 + (fq >> 1) * p.N : n0 + wave * 16 + fq * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + i * 16 + fc;
      if (m < p.M) *(float4*)(part + (size_t)m * ldp + col) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    return;
  }
  if constexpr (SILU) {
    // lanes 0-31 hold gate columns 4 (fq & 1) .. + 3, lanes 32-63 the up values of the same columns and rows
    const int n = n0 + wave * 8 + (fq & 1) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float up[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, acc[i][r]), __builtin_bit_cast(uint32_t, acc[i][r]), false, false);
        up[r] = __builtin_bit_cast(float, sw[1]);               // lanes 0-31: the value of lane + 32
      }
      const int m = m0 + i * 16 + fc;
      if (lane < 32 && m < p.M) {
        float o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // reference rounds gate/up to bf16 (Linear output) and every elementwise op after it (phi.py:469-471)
          const float g = bf16_round(acc[i][r]), u = bf16_round(up[r]);
          o4[r] = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g)))) * u;
        }
        *(u32x2_t*)((bf16_t*)p.out + (size_t)m * p.ldo + n) = (u32x2_t){pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3])};
      }
    }
    return;
  }
  const int n = n0 + wave * 16 + fq * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + i * 16 + fc;
    if (m < p.M) {
      const size_t o = (size_t)m * p.ldo + n;
      u32x2_t w = {pack_bf16x2(acc[i][0], acc[i][1]), pack_bf16x2(acc[i][2], acc[i][3])};
      if (EPI == P3V_EPI_RESID_BF16) {                         // out = resid + bf16(acc): the packed words ARE bf16(acc)
        const u32x2_t rw = *(const u32x2_t*)((const bf16_t*)p.resid + o);
#pragma unroll
        for (int k = 0; k < 2; ++k) w[k] = pack_bf16x2(bf16lo(rw[k]) + bf16lo(w[k]), bf16hi(rw[k]) + bf16hi(w[k]));
      }
      *(u32x2_t*)((bf16_t*)p.out + o) = w;
    }
  }
}

// K slices for a shape on this kernel: 0 = not one of its shapes; 1 = one pass; S > 1 = S slices + the reduction launch.
// The smallest S that gives every CU a workgroup (8 at most), every slice whole blocks of SK_D K-tiles.
int p3v_gemm_skinny_slices(int M, int N, int K, int epilogue) {
  const P3vTuning& t = p3v_tuning();
  const bool silu = epilogue == P3V_EPI_SILU_MUL;
  if (t.gemm_no_skinny || M <= 16 || M > t.gemm_skinny_max_m || N % (silu ? 32 : 64)) return 0;
  if (epilogue != P3V_EPI_NONE && epilogue != P3V_EPI_RESID_BF16 && !silu) return 0;
  const int tiles = p3v_cdiv(M, SK_BM) * ((silu ? 2 * N : N) / SK_BN);
  int best = 0;
  for (int S = 1; S <= 8; ++S) {
    if (K % (S * SK_BK * SK_D)) continue;                      // whole blocks of SK_D K-tiles per slice
    best = S;
    if (tiles * S >= t.gemm_splitk_wgs) break;
  }
  return best;
}

template <int EPI>
static int launch_skinny(const SkinnyP& p, int S, hipStream_t s) {
  static bool attr_set[2] = {false, false};
  const bool part = S > 1;
  if (!attr_set[part]) {
    const void* fn = part ? (const void*)k_gemm_skinny<EPI, true> : (const void*)k_gemm_skinny<EPI, false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS) != hipSuccess) return P3V_ERR_HIP;
    attr_set[part] = true;
  }
  const int n_tile = EPI == P3V_EPI_SILU_MUL ? SK_BN / 2 : SK_BN;
  const dim3 grid(p.N / n_tile, p3v_cdiv(p.M, SK_BM), S);
  if (part) hipLaunchKernelGGL((k_gemm_skinny<EPI, true>), grid, dim3(256), SK_LDS, s, p);
  else hipLaunchKernelGGL((k_gemm_skinny<EPI, false>), grid, dim3(256), SK_LDS, s, p);
  P3V_CHECK_LAUNCH();
  return P3V_OK;
}

int p3v_splitk_reduce(const float* part, const p3v_gemm_args_t* a, int S, hipStream_t s);   // p3v_gemm.hip

// P3V_ERR_UNSUPPORTED: not a shape for this kernel (the caller goes on to the other GEMM paths).  A split that finds no workspace of
// p3v_gemm_ws_bytes() runs as one pass.
int p3v_gemm_skinny_try(const p3v_gemm_args_t* a, hipStream_t s) {
  int S = p3v_gemm_skinny_slices(a->M, a->N, a->K, a->epilogue);
  if (S == 0) return P3V_ERR_UNSUPPORTED;
  const bool silu = a->epilogue == P3V_EPI_SILU_MUL;
  const int w_rows = silu ? 2 * a->N : a->N;
  if (S > 1 && (!a->ws || a->ws_bytes < (int64_t)S * a->M * w_rows * 4)) {
    if (a->K % (SK_BK * SK_D)) return P3V_ERR_UNSUPPORTED;
    S = 1;
  }
  if (S > 1 && ((uintptr_t)a->ws & 15)) return P3V_ERR_ARG;
  SkinnyP p = {a->A, a->W, S > 1 ? a->ws : a->out, a->resid, a->M, a->N, a->K, a->lda, a->ldw, a->ldo, S > 1 ? a->K / S : 0};
  int rc;
  if (silu) rc = launch_skinny<P3V_EPI_SILU_MUL>(p, S, s);
  else if (a->epilogue == P3V_EPI_RESID_BF16) rc = launch_skinny<P3V_EPI_RESID_BF16>(p, S, s);
  else rc = launch_skinny<P3V_EPI_NONE>(p, S, s);
  if (rc != P3V_OK || S == 1) return rc;
  return p3v_splitk_reduce((const float*)a->ws, a, S, s);
}
