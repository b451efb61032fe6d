// Interleaved prefill attention (round 3, late) -- included by p3v_attention.hip (types, AttnP and helpers come from there).
//
// What the ping-pong kernel (k_attn_prefill_pp) could not get past, measured (tools/valu_issue_rate.hip,
// tools/mfma_valu_overlap.hip, tools/gen_slot_bench.py, tools/attn_pp_timeline.py; profiles/r03_attn_il_microbench.txt):
//   * ONE wave issues a VALU instruction every ~8-10 cycles whatever the instruction (v_add, v_max3, v_exp: 4-5 ns); two or four
//     waves of a SIMD together reach 2.5 cycles (7.4 for v_exp).  A phase in which ONE wave of the SIMD does all the vector work
//     (softmax of 32 x 64 scores: ~130 instructions) is therefore ~1200 cycles long however idle the vector pipe is, and the
//     partner's 52 MFMAs (832 cycles) cannot make the step shorter: 1780 cycles per step, matrix pipe 47 % busy;
//   * VALU instructions placed BETWEEN the MFMAs of the same wave are nearly free: 26 MFMAs + 47 VALU + 12 ds_read_b128,
//     hand-interleaved, two waves per SIMD: 436-460 cycles per 26 MFMAs (416 of matrix-pipe time) -- 90 % of the matrix peak.
// So each WAVE is software-pipelined over its own two 16-query halves, half a tile apart: while the matrix pipe works on half u'
// (PV of the previous tile, then S^T of the next), the vector instructions of the softmax of half u sit between those MFMAs;
// then the roles swap.  Per tile and wave:
//   slot A(j):  this wave's DMA pieces of K(j+2), V^T(j+1), then
//               VALU softmax_1(j-1)  ||  MFMA  PV_0(j-1) , S^T_0(j)
//   -- s_barrier B_j (tiles K(j+1), V^T(j) are in LDS for everybody; the tiles read before it are free) --
//   slot B(j):  VALU softmax_0(j)    ||  MFMA  PV_1(j-1) , S^T_1(j)
//               + the fragment registers are refilled IN PLACE behind their last use: V^T(j) behind PV_1(j-1), K(j+1) behind S^T_1(j)
// Each slot is two regions (the lanes' maxima over their 16 scores || 14 MFMAs; 16 v_exp + 8 v_cvt_pk || 12 MFMAs), split by the
// rare wave-uniform branch that moves the reference (first visible tile, or a jump > 2^8).  The fast path needs NO cross-lane
// step ("some score exceeds the threshold" is a ballot over the lanes' own maxima): 9 + 24 VALU per slot, 80 VALU + 52 MFMA +
// 24 ds_read_b128 + 3 LDS-DMA + 51 SALU per tile and wave (235 instructions; ping-pong: 188 VALU + 90 SALU).  Register set
// (250-256 VGPRs, no spill), LDS fragment traffic, key-permuted K staging and the deferred reference are the ping-pong kernel's;
// ONE barrier per tile instead of two, rings of 3 + 3 tiles (72 KB at head dim 96), one counted wait (vmcnt(NPW)) from the first
// tile to the last: every wave issues its NPW pieces every iteration, tile numbers past the end refetch the last tile into a
// free slot.  The first and last tile run the same code on neutral operands (P = 0, V = 0, S = -inf) instead of peeled copies.
// Measured, B = 1, 32 heads x 96, causal (profiles/r03_attn_prefill_kernels.txt; alternated with the other kernels in one
// process): 2531 tokens 50.7-56 us with 128-query workgroups (256-query: 57-63; pp 68-76; dma 65-73), 8k 371-391 us = 1055-1111 TF/s
// (pp 430-465), 32k 5.15-5.42 ms = 1216-1281 TF/s (pp 5.83-6.17).
// Where the rest goes (SQ counters, profiles/r03_pmc_prefill_attn.txt; timing experiments P3V_IL_NO*): the matrix stream
// alone runs the 32k case in 3.84 ms, everything else alone in 3.75 ms, both in 5.39 ms -- half overlapped.  Per wave-tile
// 2890 cycles: 880 parked (barrier: the first four waves arrive ~600 cycles early every tile), 900 issue stalls, 1100 issuing.
// Q must arrive pre-scaled (scale * log2 e folded in by the RoPE kernel): the only caller is the prompt path.
#ifdef P3V_IL_DEBUG                                              // tools/attn_il_timeline.py: per-wave stamps of ONE workgroup
__device__ unsigned long long p3v_ildbg[8 * 64 * 8];
#define IL_S(k) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(il_s[k])); __builtin_amdgcn_sched_barrier(0); } while (0)
#define IL_FLUSH(it) do { if (blockIdx.x == P3V_IL_DEBUG && (threadIdx.x & 63) == 0 && (it) < 64) { for (int k_ = 0; k_ < 8; ++k_) p3v_ildbg[((threadIdx.x >> 6) * 64 + (it)) * 8 + k_] = il_s[k_]; } } while (0)
extern "C" int p3v_ildbg_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p3v_ildbg), sizeof(unsigned long long) * 8 * 64 * 8) == hipSuccess ? 0 : -1;
}
#else
#define IL_S(k) do { } while (0)
#define IL_FLUSH(it) do { } while (0)
#endif
#ifndef P3V_IL_DMA_FRONT
#define P3V_IL_DMA_FRONT 1                                      // where slot A issues the wave's DMA pieces: 1 in front of its MFMAs, 0 one every
                                                                // four MFMAs (1-2 % slower: a piece holds its wave's instruction stream ~80
                                                                // cycles wherever it stands -- inside the stream they come out of the wave's own
                                                                // MFMA issue, in front the partner wave covers them), 2 behind region 1 (= 1)
#endif
// NW = waves per workgroup: 8 (256 queries, one workgroup per CU) or 4 (128 queries, TWO workgroups per CU -- each with its own
// 72 KB of rings -- for prompts of a few thousand tokens, where 256-query blocks leave the launch bounded by its longest block:
// 2531 tokens are 10 blocks of 4..40 tiles per head, 320 workgroups for 256 CUs).  The price of NW = 4: every tile is fetched
// per 128 queries, twice the DMA pieces per wave.
template <int HD, int NW>
__global__ void __launch_bounds__(64 * NW, 8 / NW) k_attn_prefill_il(AttnP p) {
  constexpr int KROW = HD * 2, VROW = 128, NKS = HD / 32, NDT = HD / 16, CPR = HD / 8;
  constexpr int KTILE = 64 * KROW, VTILE = HD * VROW, RING = 3;
  constexpr int NK = KTILE / 1024, NV = VTILE / 1024, NPW = (NK + NV) / NW, QB = 32 * NW;   // DMA pieces per tile (K, V^T); per wave and batch
  constexpr float THR = 8.f;
  static_assert((NK + NV) % NW == 0 && (NW == 4 || NW == 8), "pieces must divide over the waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // K ring [3][KTILE] | V^T ring [3][VTILE]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, qi = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.L + QB - 1) / QB, per_group = p.head_group * nqb;
  const int hgrp = blockIdx.x / per_group, within = blockIdx.x - hgrp * per_group;
  const int qblk = nqb - 1 - within / p.head_group;             // longest (last) query blocks first
  const int b = blockIdx.z, head = hgrp * p.head_group + within % p.head_group, kvh = head / (p.nh / p.nkv);
  const int past = p.d_past ? *p.d_past : p.past;
  const int total = past + p.L;
  const int pad = p.pad_len ? p.pad_len[b / p.pad_div] : 0;
  const int qb0 = qblk * QB, q0 = qb0 + wave * 32;
  const int kv_end = p.causal ? min(total, past + qb0 + QB) : total;
  const int kv_begin = min(pad & ~63, kv_end);
  const int NT = (kv_end - kv_begin + 63) >> 6;
  const bool active = q0 < p.L;
  const int wave_last = p.causal ? past + q0 + 31 : total;  // last key position this wave can see
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

  // ---- this wave's DMA pieces: piece pi = wave + NW i of the list [K_0 .. K_{NK-1}, V_0 .. V_{NV-1}] (layout: k_attn_prefill_pp)
  const unsigned vrow = (unsigned)p.past_t * 2;
  unsigned poff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave + NW * i;
    if (pi < NK) {
      const int e = pi * 64 + lane, row = e / CPR, pc = e - row * CPR;
      const int sw = HD == 96 ? (-(row >> 2)) & 3 : (row >> 1) & 7;
      const int blk = row >> 4, i_ = row & 15;               // LDS row (block blk, A-operand row i_) holds key `key` of the tile
      const int key = 32 * (blk >> 1) + 8 * (i_ >> 2) + 4 * (blk & 1) + (i_ & 3);
      poff[i] = key * KROW + ((pc ^ sw) << 4);
    } else {
      const int e = (pi - NK) * 64 + lane, row = e >> 3, pc = e & 7;
      poff[i] = (unsigned)row * vrow + ((pc ^ ((row >> 1) & 7)) << 4);
    }
  }
  // K tile kt -> K slot kt % 3, V^T tile vt -> V slot vt % 3.  Tile numbers past the last one fetch the last tile again (the slot
  // they land in is free by construction and nobody reads it as that tile): every wave issues the same NPW pieces every
  // iteration, so ONE counted wait -- vmcnt(NPW): everything but the newest batch -- is right from the first tile to the last.
  // Per piece, wave-uniform and fixed: which tensor, where in a ring slot, and the byte stride of a tile in memory (K: 64 rows,
  // V^T: 64 columns of every row) -- a piece is then {two scalar multiply-adds, m0, one buffer_load ... lds}, no branch.
  __amdgpu_buffer_rsrc_t p_rs[NPW];
  int p_lds[NPW], p_stride[NPW], p_base[NPW], p_lag[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave + NW * i;
    const bool isk = pi < NK;
    p_rs[i] = isk ? rs_k : rs_v;
    p_lds[i] = isk ? pi * 1024 : RING * KTILE + (pi - NK) * 1024;
    p_stride[i] = isk ? 64 * KROW : 128;
    p_base[i] = isk ? kv_begin * KROW : kv_begin * 2;
    p_lag[i] = isk ? 0 : 1;                                   // a batch is K(kt) with V^T(kt - 1)
  }
  static_assert(KTILE == VTILE, "one slot size for both rings");
  // piece `i` of the batch {K(kt), V^T(kt - 1)}; slot3[0 / 1] = kt % 3, (kt - 1) % 3 (kept by the caller: no division here)
  auto issue_piece = [&](int i, int kt, int slot_k, int slot_v) __attribute__((always_inline)) {
#ifdef P3V_IL_NODMA                                              // timing experiment: tiles never move
    return;
#endif
    const int lag = p_lag[i], t = min(kt - lag, NT - 1);
    const int so = p_base[i] + t * p_stride[i];
    const unsigned vo = poff[i];                             // (a builtin called on an element of a dependent-size array: hipcc 7.2
    const __amdgpu_buffer_rsrc_t rs = p_rs[i];               //  silently drops the kernel's HOST stub -- link error, no diagnostic)
    pf_lptr_t dst = (pf_lptr_t)(smem + p_lds[i] + (lag ? slot_v : slot_k) * KTILE);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, vo, so, 0, 0);
  };
  auto issue = [&](int kt, int vt, bool with_v, int only = -1) __attribute__((always_inline)) {      // (prologue / idle waves) the whole batch, or K alone
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (with_v || wave + NW * i < NK) issue_piece(i, kt, kt % RING, (kt + RING - 1) % RING);
  };
  auto lane_now = [&]() __attribute__((always_inline)) {                                    // a lane id the compiler cannot see through (k_attn_prefill_pp)
    unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };

  const u32x4_t ones_w = qi == 0 ? (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : (u32x4_t){0u, 0u, 0u, 0u};
  const bf16x8_t ones_f = __builtin_bit_cast(bf16x8_t, ones_w);
  const bf16x8_t zero_f = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
  float m_run[2] = {0.f, 0.f};                               // reference of the exponent (finite always)
  unsigned long long unset[2] = {~0ull, ~0ull};              // lane mask: no visible key seen yet, the next one sets the reference
                                                             // (a MASK, not a per-lane bool: the fast path's test is then one v_cmp and
                                                             //  one s_or; as a bool the ballot is re-materialised with two more VALU)
  f32x4_t ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4_t o[2][NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < NDT; ++d) o[u][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t s[2][4];
  bf16x8_t pf[2][2];
#ifdef P3V_IL_DEBUG
  unsigned long long il_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  f32x4_t negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // C operand of the S^T products: -reference
  bf16x8_t kfr[4][NKS];                                      // K fragments of tile j     (refilled in place in slot B)
  bf16x8_t vfr[2][NDT];                                      // V^T fragments of tile j-1 (refilled in place in slot B)
  // neutral operands of the first iteration: softmax_1(-1) sees S = -inf (P = 0, reference untouched), PV(-1) = 0 . 0
#pragma unroll
  for (int st = 0; st < 4; ++st) s[1][st] = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    pf[0][st] = zero_f;
#pragma unroll
    for (int d = 0; d < NDT; ++d) vfr[st][d] = zero_f;
  }

  // ---- pieces of a slot.  u = the half whose softmax is in work (VALU), uu = 1 - u the half on the matrix pipe.
  auto mask_half = [&](int u, int j) __attribute__((always_inline)) {                      // (rare) tile j of half u has invisible keys: -inf in place
    const int kv0 = kv_begin + 64 * j;
    const float ninf = -INFINITY;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = kv0 + 32 * (st >> 1) + 8 * g + 4 * (st & 1) + r;      // key order of the staged tile
        const bool vis = t < kv_end && t >= pad && (!p.causal || t <= qpos[u]) && qpos[u] >= pad;
        const unsigned long long vm = __builtin_amdgcn_ballot_w64(vis);
        asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(s[u][st][r]) : "v"(ninf), "s"(vm));      // s = vis ? s : -inf
      }
  };
  auto move_reference = [&](int u, float m_t) __attribute__((always_inline)) {             // (rare, wave-uniform) in-place: see k_attn_prefill_pp
    const bool masked = m_t == -INFINITY;
    const bool un = (unset[u] >> lane_now()) & 1;
    const float delta = un ? (masked ? 0.f : m_t) : fmaxf(m_t, 0.f);
    const float alpha = un ? 1.f : __builtin_amdgcn_exp2f(-delta);
    unset[u] &= __builtin_amdgcn_ballot_w64(masked);
    m_run[u] += delta;
    const float nm = -m_run[u];                              // (in place: a new value of negm costs the FAST path four register copies per slot)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm("v_mov_b32 %0, %1" : "+v"(negm[u][r]) : "v"(nm));
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
  };
  // ---- one slot: the softmax of (half u, tile j) BETWEEN the MFMAs of PV_uu (tile in vfr) and S^T_uu (tile in kfr).  The
  // instruction order is the source order: every group {one MFMA, its fragment refill, two or three VALU} ends in a
  // sched_barrier (nothing crosses), the VALU stream is written one machine instruction per statement.  (Prescribing the same
  // order with sched_group_barrier worked for the maxima and failed for the exponentials / the refills: 12 v_exp ahead of the
  // first MFMA, 9 ds_read behind the last.)
  unsigned voff[2], koff[NKS];                              // fragment read offsets inside a tile: lane constants, five registers kept
  {                                                         // live across the loop (recomputed per slot: 15 VALU more per tile, -2.5 %)
    voff[0] = qi * VROW + (((0 + g) ^ ((qi >> 1) & 7)) << 4);
    voff[1] = qi * VROW + (((4 + g) ^ ((qi >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const unsigned c = 4 * ks + g, sw = HD == 96 ? (0u - (qi >> 2)) & 3 : (qi >> 1) & 7;
      koff[ks] = qi * KROW + ((c ^ sw) << 4);
    }
  }
  auto slot = [&](int u, int j, bool refill, const unsigned char* Vn, const unsigned char* Kn, int dma_kt = -1, int dma_sk = 0, int dma_sv = 0) __attribute__((always_inline)) {
    const int uu = 1 - u;
    const bool interior = j >= j_int_lo && j <= j_int_hi;
    if (!interior) mask_half(u, j);
    __builtin_amdgcn_sched_barrier(0);
#if P3V_IL_DMA_FRONT == 1
    if (dma_kt >= 0) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) issue_piece(i, dma_kt, dma_sk, dma_sv);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // region 1: PV_uu (2 NDT + 2 MFMAs) || maximum of the 64 scores of each query of half u
    // (the fast path needs no cross-lane step: "some score of the wave exceeds the threshold" is a ballot over the LANES' own
    //  maxima; the row maximum is only formed when the reference does move)
    float ma = 0.f, mc = 0.f;
    auto max3 = [](float x, float y, float z) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; };
    auto max2 = [](float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto valu1 = [&](int k) __attribute__((always_inline)) {                                // step k of the reduction: one or two instructions
      switch (k) {
        case 0: ma = max3(s[u][0][0], s[u][0][1], s[u][0][2]); mc = max3(s[u][0][3], s[u][1][0], s[u][1][1]); break;
        case 1: ma = max3(ma, s[u][1][2], s[u][1][3]); mc = max3(mc, s[u][2][0], s[u][2][1]); break;
        case 2: ma = max3(ma, s[u][2][2], s[u][2][3]); mc = max3(mc, s[u][3][0], s[u][3][1]); break;
        case 3: ma = max3(ma, s[u][3][2], s[u][3][3]); break;
        case 4: ma = max2(ma, mc); break;
        default: break;
      }
    };
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int d = 0; d <= NDT; ++d) {
        if (d < NDT) {
#ifdef P3V_IL_NOMFMA                                             // timing experiment: no matrix work (an in-place no-op keeps the operands used)
          asm volatile("" : "+v"(o[uu][d]) : "v"(vfr[st][d]), "v"(pf[uu][st]));
#else
          o[uu][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[st][d], pf[uu][st], o[uu][d], 0, 0, 0);
#endif
#ifndef P3V_IL_NOLDS                                             // (timing experiment: no fragment refills)
          if (refill) vfr[st][d] = *(const bf16x8_t*)(Vn + voff[st] + d * 16 * VROW);
#endif
        } else {
#ifndef P3V_IL_NOMFMA
          ol[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[uu][st], ol[uu], 0, 0, 0);
#endif
        }
#ifndef P3V_IL_NOVALU                                            // (timing experiment: no softmax work)
        valu1(st * (NDT + 1) + d);
#endif
        if (dma_kt >= 0) {                // this wave's DMA pieces, one at a time between MFMAs (a burst of three
          const int gi = st * (NDT + 1) + d;                 // right after the barrier, all eight waves at once, cost each wave ~300 cycles)
#if P3V_IL_DMA_FRONT == 0
          if (gi % 4 == 1 && gi / 4 < NPW) issue_piece(gi / 4, dma_kt, dma_sk, dma_sv);
#elif P3V_IL_DMA_FRONT == 2
          if (gi >= 2 * NDT + 2 - NPW) issue_piece(gi - (2 * NDT + 2 - NPW), dma_kt, dma_sk, dma_sv);   // behind the region's last MFMAs
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    IL_S(u ? 1 : 5);
    if ((unset[u] | __builtin_amdgcn_ballot_w64(ma > THR)) != 0) move_reference(u, rows_max(ma));
    __builtin_amdgcn_sched_barrier(0);
    // region 2: S^T_uu (4 NKS MFMAs; k-slice outermost: four independent accumulators between two MFMAs of a chain) ||
    // P = 2^S of half u (16 v_exp) and its packing into the B fragments of the PV product (8 v_cvt_pk)
    // (the machine code follows this order only roughly -- pure instructions are placed before the sched_barriers exist:
    //  v_exp in fours, the 8 v_cvt_pk behind the last MFMA.  Forcing the exact order with data ties (empty asm taking one
    //  instruction's result and the next one's operand) was measured 1.3 % SLOWER: not the lever)
    u32x4_t pw[2];
    constexpr int NG = 4 * NKS, NE = NG >= 12 ? 8 : 4, EPG = 16 / NE;        // groups that carry exponentials; exponentials per group
    constexpr int NC = NG - NE >= 4 ? 4 : NG - NE, CPG = 8 / NC;             // packing groups; v_cvt_pk per group
    static_assert(EPG == 2 || EPG == 4, "two or four exponentials per group");
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int k = 4 * ks + st;
        if (u == 1 && k == 4) IL_S(4);
        if (u == 1 && k == 8) IL_S(7);
#ifdef P3V_IL_NOMFMA
        asm volatile("" : "+v"(s[uu][st]) : "v"(kfr[st][ks]), "v"(qf[uu][ks]));
#else
        s[uu][st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[st][ks], qf[uu][ks], ks == 0 ? negm[uu] : s[uu][st], 0, 0, 0);
#endif
#ifndef P3V_IL_NOLDS
        if (refill) kfr[st][ks] = *(const bf16x8_t*)(Kn + koff[ks] + st * 16 * KROW);
#endif
#ifdef P3V_IL_NOVALU
        if (false) {
#else
        if (k < NE) {
#endif
          const int eb = (EPG * k) >> 2, e0 = (EPG * k) & 3;                   // EPG consecutive scores (same accumulator block)
#pragma unroll
          for (int e = e0; e < e0 + EPG; ++e) {
#ifdef P3V_IL_NOEXP                                              // timing experiment: a plain VALU op in place of the transcendental
            asm("v_mul_f32 %0, 0.5, %0" : "+v"(s[u][eb][e]));
#else
            s[u][eb][e] = __builtin_amdgcn_exp2f(s[u][eb][e]);
#endif
          }
#ifdef P3V_IL_NOVALU
        } else if (false) {
#else
        } else if (k < NE + NC) {
#endif
#pragma unroll
          for (int c = CPG * (k - NE); c < CPG * (k - NE + 1); ++c) {         // word c of [st' = c >> 2][c & 3]
            const int st2 = c >> 2, w = c & 3, blk = 2 * st2 + (w >> 1), r0 = 2 * (w & 1);
            pw[st2][w] = pack_bf16x2(s[u][blk][r0], s[u][blk][r0 + 1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#ifndef P3V_IL_NOVALU
    pf[u][0] = __builtin_bit_cast(bf16x8_t, pw[0]);
    pf[u][1] = __builtin_bit_cast(bf16x8_t, pw[1]);
#endif
    // P and S^T are USED here as far as the optimiser can tell (their real uses sit behind the barrier / in the next slot:
    // without this the exponentials are sunk there, out of the MFMA stream they are meant to hide in)
    asm volatile("" ::"v"(pf[u][0]), "v"(pf[u][1]), "v"(s[uu][0]), "v"(s[uu][1]), "v"(s[uu][2]), "v"(s[uu][3]));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto pv_tail = [&](int uu) __attribute__((always_inline)) {                               // PV of half uu alone (the last tile's second half)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) o[uu][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[st][d], pf[uu][st], o[uu][d], 0, 0, 0);
      ol[uu] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_f, pf[uu][st], ol[uu], 0, 0, 0);
    }
  };
  auto step_barrier = [&]() __attribute__((always_inline)) {                               // B_j: everything but this wave's newest batch has landed
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  const int NTw = active ? max(0, min(NT, ((wave_last - kv_begin) >> 6) + 1)) : 0;   // tiles this wave works on (0 .. NTw-1)
  if (NT > 0) {
    issue(0, 0, false);                                      // K(0) alone
    issue(1, 0, true);                                       // batch 0 = K(1), V^T(0)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");   // K(0) is in LDS, batch 0 may still fly
    asm volatile("s_barrier" ::: "memory");
    {
      const unsigned ln = lane_now(), g_ = ln >> 4, qi_ = ln & 15;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const unsigned c = 4 * ks + g_, sw = HD == 96 ? (0u - (qi_ >> 2)) & 3 : (qi_ >> 1) & 7;
        const unsigned off = qi_ * KROW + ((c ^ sw) << 4);
#pragma unroll
        for (int st = 0; st < 4; ++st) kfr[st][ks] = *(const bf16x8_t*)(smem + off + st * 16 * KROW);
      }
    }
    // (Measured and dropped: waves 4-7 half an iteration BEHIND waves 0-3 -- their barrier in front of slot A instead of behind
    //  it, so that every wave runs one slot A and one slot B between two barriers and nobody parks at the barrier -- 1-2 %
    //  slower; two copies of the loop for it cost 46 spilled registers.)
    {
      int j = 0, r0 = 0, r1 = 1, r2 = 2;                     // j % 3, (j + 1) % 3, (j + 2) % 3
      for (; j < NTw; ++j) {
        IL_S(0);
        // A(j): softmax_1(j-1) || PV_0(j-1), S^T_0(j)   + this wave's DMA pieces of K(j+2), V^T(j+1)
        slot(1, j - 1, false, nullptr, nullptr, j + 2, r2, r1);
        IL_S(2);
        step_barrier();
        IL_S(3);
        slot(0, j, true, smem + RING * KTILE + r0 * VTILE, smem + r1 * KTILE);   // B(j): softmax_0(j) || PV_1(j-1), S^T_1(j)
        IL_S(6);
        IL_FLUSH(j);
        const int t_ = r0;
        r0 = r1, r1 = r2, r2 = t_;
      }
      if (active) {                                          // j = NTw: what is left of the last tile: softmax_1, PV_0, then PV_1
        slot(1, j - 1, false, nullptr, nullptr, j + 2, r2, r1);   // (its S^T_0 runs on the refetched last tile; nobody reads the result)
        step_barrier();
        pv_tail(1);
        ++j;
      }
      for (; j <= NT; ++j) {                                 // done (or never had rows): DMA share + barriers only
        issue(j + 2, 0, true);
        step_barrier();
      }
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
