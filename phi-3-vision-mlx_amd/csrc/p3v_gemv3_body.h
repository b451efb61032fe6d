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

// wpw: waves of the 4-wave workgroup that take rows (4, or 3: wave 3 then only helps with the prologue).  1536 streaming waves
// (qkv, o_proj, down) as 384 four-wave workgroups put two workgroups on half of the CUs and one on the others, and the launch
// lasts as long as the doubly loaded CUs; 512 workgroups x 3 waves load every CU alike (tools/gemv_timeline.py).
template <int MT, int NST, int CH>
__device__ __forceinline__ void gemv3_body(const GemvP& p, int units_per_wave, int bx, unsigned char* smem, float* red, int wpw = 4) {
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
#pragma unroll
  for (int k = 0; k < XC; ++k) {
    const int c = min(tid + k * 256, CHUNKS - 1);
#pragma unroll
    for (int m = 0; m < MT; ++m) xv[m][k] = ((const u32x4_t*)(p.x + (size_t)min(m, p.M - 1) * (CHUNKS * 8)))[c];
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
}
