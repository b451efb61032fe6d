// Body of the streaming M = 1 GEMV (k_gemv3, p3v_gemv.hip), shared with the o_proj stage of the fused attention launch
// (fo_project, p3v_attention.hip).  (Two more fused launches once shared it -- a GEMV chain and a qkv-projection + attention launch,
// both bit-exact, both slower than separate launches; removed in round 4, described in DESIGN.md section 3.1.)
#pragma once
#include <type_traits>

#include "p3v_common.h"

struct GemvP {
  const bf16_t* x; const bf16_t* W; void* out; const bf16_t* resid; const bf16_t* norm_w;
  float eps;
  int M, N, K, epi, units;
};

// 8 bf16 x 8 bf16 -> fp32 accumulate on v_dot2c_f32_bf16 (two products per instruction straight from the packed
// operands): 4 VALU instructions per 16-byte weight chunk instead of 8 unpacks + 8 FMAs -- the GEMV's VALU pipe was
// ~60 % busy with unpacking before.
// (The pairs are taken with shufflevector from an 8 x bf16 view: hipcc 7.2 folds `bit_cast<2 x bf16>(w[j])` of the four
// dwords of a u32x4 into element 0 -- same family of bug as the permlane-swap fold noted in p3v_common.h.)
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16oct_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float dot8(u32x4_t w, u32x4_t x, float acc) {
  const bf16oct_t wv = __builtin_bit_cast(bf16oct_t, w), xv = __builtin_bit_cast(bf16oct_t, x);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(xv, xv, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(xv, xv, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(xv, xv, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(xv, xv, 6, 7), acc, false);
  return acc;
}

typedef std::integral_constant<int, 0> IC0;
typedef std::integral_constant<int, 1> IC1;

// Round 6: the two ends of a replayed greedy decode step ride in the step's first and last projection (p3v_gemv_step):
//  STEP_BEGIN  x row m = embed_table[clamp(tok[m])] (what p3v_step_begin gathered into x_out); workgroup 0 also writes the rows to
//              x_out -- the residual stream the later launches update in place -- and stages the rotation rows of position *d_past.
//  STEP_END    arg-max of the output rows (first maximum of the bf16 values; a NaN row reports -1) + p3v_step_end's bookkeeping:
//              every workgroup publishes its (value, index) candidates write-through and takes a ticket, the LAST one reduces them.
enum { STEP_NONE = 0, STEP_BEGIN = 1, STEP_END = 2 };
struct GemvStepP {
  const int32_t* tok; const bf16_t* table; int vocab; bf16_t* x_out;
  const float* cos_t; const float* sin_t; const int32_t* d_past_in; float* cos_o; float* sin_o; int tab_t, half;
  int32_t* next_tok; int32_t* tok_out; int32_t* hist; int32_t* d_step; int32_t* d_past; int32_t* ticket; float* amax_ws; int max_steps;
};
struct ArgMaxVI { float v; int i; };
__device__ __forceinline__ ArgMaxVI amax_better(ArgMaxVI a, ArgMaxVI b) {   // larger value, then smaller index (k_argmax's order)
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ void amax_take(ArgMaxVI& m, float v, int i) {
  if (v != v) { v = INFINITY; i = -1; }                         // a NaN logit: (+inf, -1) beats every real entry (api._rows raises)
  if (v > m.v || (v == m.v && i < m.i) || m.i == 0x7fffffff) { m.v = v; m.i = i; }
}

// ---- the two ends of a replayed greedy step, carried by its first / last projection (p3v_gemv_step, p3v_gemv_fp8_step):
// 5a. (workgroup 0 of the first qkv launch) the gathered embedding rows become the residual stream the later launches update in place
// (re-read from the table: L2-hot, and nothing has to keep them in registers), and the rotation rows of position *d_past are staged for
// the attention launches -- what p3v_step_begin did
template <int MT>
__device__ __forceinline__ void gemv_step_begin_tail(const GemvStepP* sp, const bf16_t* const* xrow, int M, int chunks, int tid) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
    if (m < M)
      for (int c = tid; c < chunks; c += 256) ((u32x4_t*)(sp->x_out + (size_t)m * (chunks * 8)))[c] = ((const u32x4_t*)xrow[m])[c];
  const int past = *sp->d_past_in;
  for (int i = tid; i < M * sp->half; i += 256) {
    const int b = i / sp->half, d = i - b * sp->half;
    sp->cos_o[i] = sp->cos_t[((size_t)b * sp->tab_t + past) * sp->half + d];
    sp->sin_o[i] = sp->sin_t[((size_t)b * sp->tab_t + past) * sp->half + d];
  }
}

// 5b. (every workgroup of the vocabulary head) arg-max + loop bookkeeping without a launch of their own -- what p3v_argmax +
// p3v_step_end did: `best` = each wave's candidates (lane 0's copy counts); the last workgroup to finish reduces all of them
template <int MT>
__device__ __forceinline__ void gemv_step_end_tail(const GemvStepP* sp, const ArgMaxVI (&best)[MT], int M, int bx, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ ArgMaxVI wbest[4][MT];
  __shared__ int s_last;
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) wbest[wave][m] = best[m];
  }
  __syncthreads();
  const int n_wg = gridDim.x;
  if (tid == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      ArgMaxVI r = wbest[0][m];
#pragma unroll
      for (int w = 1; w < 4; ++w) r = amax_better(r, wbest[w][m]);
      // one 8-byte write-through store per row: a (value, index) pair is never seen half-written by the reducer below
      const unsigned long long rec = (unsigned long long)__builtin_bit_cast(uint32_t, r.v) | ((unsigned long long)(uint32_t)r.i << 32);
      __hip_atomic_store((unsigned long long*)sp->amax_ws + ((size_t)bx * MT + m), rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the candidates are out before this workgroup counts itself in
    // Arrival in two levels (MI355X_MICROARCH.md, fanin: ~12 ns per atomic on one word -- the 1002 workgroups of the e4m3 vocabulary head
    // finish together and queued for ~10 us on a single ticket): workgroup b counts on counter b % 8 (b % 8 is also its XCD), the last of
    // each group on the top counter; the last of those has seen everything.  Counters: behind the records, 128 bytes apart, left zero.
    int* ctr = (int*)((unsigned long long*)sp->amax_ws + P3V_GEMV_STEP_MAX_WG * MT);
    const int grp = bx & 7, n_grp = min(8, n_wg), in_grp = (n_wg - grp + 7) >> 3;
    int last = 0;
    if (__hip_atomic_fetch_add(ctr + 32 * grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1)
      last = __hip_atomic_fetch_add(ctr + 32 * 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grp - 1;
    s_last = last;
  }
  __syncthreads();
  if (s_last) {                                              // the last workgroup to finish: every other candidate is visible
    __shared__ ArgMaxVI fin[4][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      ArgMaxVI r = ArgMaxVI{-INFINITY, 0x7fffffff};
      for (int g = tid; g < n_wg; g += 256) {
        const unsigned long long rec = __hip_atomic_load((const unsigned long long*)sp->amax_ws + ((size_t)g * MT + m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = amax_better(r, ArgMaxVI{__builtin_bit_cast(float, (uint32_t)rec), (int)(uint32_t)(rec >> 32)});
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        ArgMaxVI o;
        o.v = __shfl_xor(r.v, off, 64);
        o.i = __shfl_xor(r.i, off, 64);
        r = amax_better(r, o);
      }
      if (lane == 0) fin[wave][m] = r;
    }
    __syncthreads();
    if (tid == 0) {
      const int step = *sp->d_step, past_now = *sp->d_past;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (m < M) {
          ArgMaxVI r = fin[0][m];
#pragma unroll
          for (int w = 1; w < 4; ++w) r = amax_better(r, fin[w][m]);
          const int idx = r.i == 0x7fffffff ? 0 : r.i;
          sp->next_tok[m] = idx;
          sp->tok_out[m] = idx;
          if (step < sp->max_steps) sp->hist[(size_t)m * sp->max_steps + step] = idx;
        }
      *sp->d_step = step + 1;
      *sp->d_past = past_now + 1;
      int* ctr = (int*)((unsigned long long*)sp->amax_ws + P3V_GEMV_STEP_MAX_WG * MT);
      for (int c = 0; c < 9; ++c) __hip_atomic_store(ctr + 32 * c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next replay
    }
  }
}

// wpw: waves of the 4-wave workgroup that take rows (4, or 3: wave 3 then only helps with the prologue).  1536 streaming waves
// (qkv, o_proj, down) as 384 four-wave workgroups put two workgroups on half of the CUs and one on the others, and the launch
// lasts as long as the doubly loaded CUs; 512 workgroups x 3 waves load every CU alike (tools/gemv_timeline.py).
template <int MT, int NST, int CH, int STEP = STEP_NONE>
__device__ __forceinline__ void gemv3_body(const GemvP& p, int units_per_wave, int bx, unsigned char* smem, float* red, int wpw = 4,
                                           const GemvStepP* sp = nullptr) {
  constexpr int CHUNKS = NST * CH * 64;                 // 16-byte chunks per row (K = 8 * CHUNKS)
  constexpr int XC = (CHUNKS + 255) / 256;              // x chunks per thread
  u32x4_t* xs = (u32x4_t*)smem;                         // [MT][CHUNKS] bf16 x (normalised)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool silu = p.epi == P3V_EPI_SILU_MUL;
  const bool has_res = p.epi == P3V_EPI_RESID_BF16;
  const int u_begin = wave < wpw ? min(p.units, (bx * wpw + wave) * units_per_wave) : p.units;
  const int u_end = min(p.units, u_begin + units_per_wave);
  const int n_st = (u_end - u_begin) * NST;

  // ---- 1. x / norm-weight loads (oldest in the queue)
  u32x4_t xv[MT][XC], gv[XC];
  const bf16_t* xrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (STEP == STEP_BEGIN) {
      int id = sp->tok[min(m, p.M - 1)];                       // (uniform: a scalar load)
      id = id < 0 ? 0 : (id >= sp->vocab ? sp->vocab - 1 : id);
      xrow[m] = sp->table + (size_t)id * (CHUNKS * 8);
    } else {
      xrow[m] = p.x + (size_t)min(m, p.M - 1) * (CHUNKS * 8);
    }
  }
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = min(tid + k * 256, CHUNKS - 1);
#pragma unroll
    for (int m = 0; m < MT; ++m) xv[m][k] = ((const u32x4_t*)xrow[m])[c];
    gv[k] = p.norm_w ? ((const u32x4_t*)p.norm_w)[c] : (u32x4_t){0, 0, 0, 0};
  }

  // ---- 2. weight pipeline state
  u32x4_t wbuf[2][2][CH];
  uint32_t rbuf[2][MT];                                  // residual pair (2 bf16) of the row pair, per x row
  auto issue = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int u = min(u_begin + gs / NST, p.units - 1), s = gs % NST;
    const int r0 = silu ? u : 2 * u;
    const int r1 = silu ? u + p.N : min(2 * u + 1, p.N - 1);
    const u32x4_t* w0 = (const u32x4_t*)(p.W + (size_t)r0 * (CHUNKS * 8)) + s * CH * 64 + lane;
    const u32x4_t* w1 = (const u32x4_t*)(p.W + (size_t)r1 * (CHUNKS * 8)) + s * CH * 64 + lane;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      wbuf[buf][0][j] = __builtin_nontemporal_load(w0 + j * 64);
      wbuf[buf][1][j] = __builtin_nontemporal_load(w1 + j * 64);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {                       // 4-byte aligned: r0 = 2u is even, N is even on this path
      const bf16_t* rp = p.resid + (size_t)min(m, p.M - 1) * p.N + 2 * u;
      rbuf[buf][m] = !has_res ? 0u : *(const uint32_t*)rp;
    }
  };
  if (n_st > 0) issue(0, IC0{});

  // ---- 3. RMSNorm prologue (waits for the x loads only) -> LDS
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float r = 1.f;
    if (p.norm_w) {
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < XC; ++k) {
        if (tid + k * 256 < CHUNKS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float a = bf16lo(xv[m][k][j]), b = bf16hi(xv[m][k][j]); ss += a * a + b * b; }
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) red[wave + 4 * (m & 1)] = ss;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const float* rr = red + 4 * (m & 1);
      r = rsqrtf(((rr[0] + rr[1]) + (rr[2] + rr[3])) / (float)(CHUNKS * 8) + p.eps);
    }
#pragma unroll
    for (int k = 0; k < XC; ++k) {
      const int c = tid + k * 256;
      if (c < CHUNKS) {
        u32x4_t o = xv[m][k];
        if (p.norm_w) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = rms_pair(xv[m][k][j], r, gv[k][j]);
        }
        xs[m * CHUNKS + c] = m < p.M ? o : (u32x4_t){0, 0, 0, 0};
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- 4. pipeline
  float a0[MT], a1[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a0[m] = a1[m] = 0.f;
  ArgMaxVI best[MT];                                         // STEP_END: this wave's arg-max candidates (lane 0's copy counts)
#pragma unroll
  for (int m = 0; m < MT; ++m) best[m] = ArgMaxVI{-INFINITY, 0x7fffffff};
  auto compute = [&](int gs, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int s = gs % NST;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x4_t xa = xs[m * CHUNKS + (s * CH + j) * 64 + lane];
        a0[m] = dot8(wbuf[buf][0][j], xa, a0[m]);
        a1[m] = dot8(wbuf[buf][1][j], xa, a1[m]);
      }
    }
    if (s == NST - 1) {                                   // row pair complete (compile-time true when NST == 1)
      const int u = u_begin + gs / NST;
#pragma unroll
      for (int m = 0; m < MT; ++m) { a0[m] = wave_sum(a0[m]); a1[m] = wave_sum(a1[m]); }
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (m < p.M) {
            if (silu) {
              const float g = bf16_round(a0[m]), up = bf16_round(a1[m]);
              const float sg = bf16_round(g * bf16_round(1.f / (1.f + __expf(-g))));
              bf16_t* dst = (bf16_t*)p.out + (size_t)m * p.N + u;
              *dst = f32_to_bf16(sg * up);
            } else if (p.epi == P3V_EPI_F32) {
              ((float*)p.out)[(size_t)m * p.N + 2 * u] = a0[m];
              ((float*)p.out)[(size_t)m * p.N + 2 * u + 1] = a1[m];
            } else {
              float v0 = a0[m], v1 = a1[m];
              if (has_res) { v0 = bf16lo(rbuf[buf][m]) + bf16_round(v0); v1 = bf16hi(rbuf[buf][m]) + bf16_round(v1); }
              uint32_t* dst = (uint32_t*)((bf16_t*)p.out + (size_t)m * p.N + 2 * u);
              *dst = pack_bf16x2(v0, v1);
              if (STEP == STEP_END) {                          // on the values just stored (bf16)
                amax_take(best[m], bf16_round(v0), 2 * u);
                if (2 * u + 1 < p.N) amax_take(best[m], bf16_round(v1), 2 * u + 1);
              }
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) a0[m] = a1[m] = 0.f;
    }
  };
  int gs = 0;
  while (gs + 2 < n_st) {                                 // branch-free body: counted vmcnt survives
    issue(gs + 1, IC1{});
    compute(gs, IC0{});
    issue(gs + 2, IC0{});
    compute(gs + 1, IC1{});
    gs += 2;
  }
  if (gs + 1 < n_st) {
    issue(gs + 1, IC1{});
    compute(gs, IC0{});
    compute(gs + 1, IC1{});
  } else if (gs < n_st) {
    compute(gs, IC0{});
  }
  if (STEP == STEP_BEGIN && bx == 0) gemv_step_begin_tail<MT>(sp, xrow, p.M, CHUNKS, tid);
  if (STEP == STEP_END) gemv_step_end_tail<MT>(sp, best, p.M, bx, tid);
}
