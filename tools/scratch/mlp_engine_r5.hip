// Round 5 probe: the decode MLP pair (RMSNorm-less gate_up + SiLU*up -> down + residual, M = 1, Phi-3 shapes: 151 MB of weights per
// layer) as ONE PERSISTENT LAUNCH on the loader / consumer engine MI355X_MICROARCH.md describes (rows ldsdma-fill, allgather,
// prefetch-credit, engine-vs-launches): per CU one LDS-DMA LOADER wave that streams the CU's weight rows through a ring of 8 x 16 KiB
// (non-temporal) and never waits for a dependency edge -- only for ring space --, three CONSUMER waves on v_dot2c that retire ring
// slots, and hand-overs as data-tagged 4-byte granules {bf16 value, 16-bit generation}: the producers store them write-through, one
// wave per CU sweeps the whole vector (32 KB for `a`, 12 KB for `x`) into LDS and re-reads what has not arrived.  NL layers are
// chained in the launch, so both edges of the pair (a: gate_up -> down, x: down -> the next gate_up) are inside the timed loop.
// Against it: the product's two launches per pair (k_gemv3: 17.0 + 9.8 us in the step's kernel trace).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../phi-3-vision-mlx_amd/csrc -o mlp_engine_r5 mlp_engine_r5.hip && ./mlp_engine_r5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "p3v_common.h"
#include "p3v_gemv3_body.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int H = 3072, I = 8192, NCU = 256, NWSET = 8;
constexpr int GU_SLOTS = I / NCU;          // 32: slot = gate row + up row of one output column (12 KiB)
constexpr int DN_ROWS = H / NCU;           // 12: slot = one down_proj row (16 KiB)
constexpr int SLOTS_PER_LAYER = GU_SLOTS + DN_ROWS;
constexpr int RING = 8, SLOT_BYTES = 16384;
#ifndef NLOAD
#define NLOAD 2                       // loader waves per CU (one issues ~1 KiB per 150 cycles = 16 GB/s: 4.1 TB/s over the chip)
#endif
#ifndef DEPTH
#define DEPTH 3                       // fills in flight behind the one being issued (vmcnt counts at most 63 instructions)
#endif
constexpr int LDS_BYTES = RING * SLOT_BYTES + H * 2 + I * 2 + 256;

struct EngP {
  const bf16_t* Wgu[NWSET]; const bf16_t* Wd[NWSET];   // [2 I, H], [H, I]
  uint32_t* A[2]; uint32_t* X[2];                      // granules {bf16 value | tag << 16}: a [I], x [H]
  int n_layers;
  int mode;                                            // experiments: 1 = no gathers (stale vectors), 2 = consumers only free their slots
  long long* stamps;                                   // optional: per-CU wall-clock marks [cu][8] of layer `stamp_layer`
  int stamp_layer;
};

__device__ __forceinline__ void st_gran(uint32_t* p, float v, int tag) {
  __hip_atomic_store(p, (uint32_t)f32_to_bf16(v) | ((uint32_t)tag << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int lds_poll(volatile int* p) { return *p; }

// one wave sweeps a granule vector of n words (a multiple of 256) into LDS as packed bf16: every 16-byte load of the sweep in flight at
// once (agent-scope: sc1), the instructions whose granules have not all arrived are read again
template <int N_INST>
__device__ __forceinline__ void gather(const uint32_t* g, bf16_t* dst, int tag, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0xffffffff, 0x00020000);
  u32x4_t w[N_INST];
  unsigned long long pending = N_INST == 64 ? ~0ull : ((1ull << N_INST) - 1);
  const uint32_t t = (uint32_t)tag;
  while (pending) {
#pragma unroll
    for (int i = 0; i < N_INST; ++i)
      if ((pending >> i) & 1) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 64 + lane) * 16, 0, 16);
#pragma unroll
    for (int i = 0; i < N_INST; ++i)
      if ((pending >> i) & 1) {
        const bool ok = (w[i][0] >> 16) == t && (w[i][1] >> 16) == t && (w[i][2] >> 16) == t && (w[i][3] >> 16) == t;
        if (__all(ok)) {
          *(u32x2_t*)(dst + (i * 64 + lane) * 4) = (u32x2_t){(w[i][0] & 0xffffu) | (w[i][1] << 16), (w[i][2] & 0xffffu) | (w[i][3] << 16)};
          pending &= ~(1ull << i);
        }
      }
    if (pending) __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ void wait_vm(int n) {             // s_waitcnt vmcnt(n), n a multiple of 4 up to 60
  switch (n >> 2) {
#define VMC(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * k) : "memory"); break;
    VMC(0) VMC(1) VMC(2) VMC(3) VMC(4) VMC(5) VMC(6) VMC(7) VMC(8) VMC(9) VMC(10) VMC(11) VMC(12) VMC(13) VMC(14) VMC(15)
#undef VMC
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
  }
}

__global__ void __launch_bounds__(64 * (NLOAD + 3), 1) k_mlp_engine(EngP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  bf16_t* xs = (bf16_t*)(smem + RING * SLOT_BYTES);
  bf16_t* as = xs + H;
  volatile int* ready = (volatile int*)(as + I);        // [RING] generation landed
  volatile int* freed = ready + RING;                   // [RING] generation consumed
  volatile int* vec_gen = freed + RING;                 // [0]: x generation in LDS, [1]: a generation in LDS
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cu = blockIdx.x;
  if (tid < 2 * RING + 2) ((int*)ready)[tid] = 0;
  __syncthreads();
  const int total = p.n_layers * SLOTS_PER_LAYER;

  if (wave < NLOAD) {
    // ---- LOADERS: loader w fills the global slots g = w (mod NLOAD) in order, DEPTH fills in flight behind the one being issued
    int hist[DEPTH] = {};
    for (int g = wave; g < total + DEPTH * NLOAD; g += NLOAD) {
      int n_inst = 0;
      if (g < total) {
        const int L = g / SLOTS_PER_LAYER, s = g % SLOTS_PER_LAYER, slot = g % RING, gen = g / RING;
        while (lds_poll(&freed[slot]) != gen) __builtin_amdgcn_s_sleep(1);          // ring space (never a dependency edge)
        unsigned char* dst = ring + slot * SLOT_BYTES;
        if (s < GU_SLOTS) {
          const bf16_t* W = p.Wgu[L % NWSET];
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffff, 0x00020000);
          const unsigned r0 = (unsigned)(cu * GU_SLOTS + s) * (H * 2), r1 = (unsigned)(I + cu * GU_SLOTS + s) * (H * 2);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 1024), 16, r0 + lane * 16, i * 1024, 0, 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + 6144 + i * 1024), 16, r1 + lane * 16, i * 1024, 0, 2);
          }
          n_inst = 12;
        } else {
          const bf16_t* W = p.Wd[L % NWSET];
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffff, 0x00020000);
          const unsigned r0 = (unsigned)(cu * DN_ROWS + (s - GU_SLOTS)) * (I * 2);
#pragma unroll
          for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 1024), 16, r0 + lane * 16, i * 1024, 0, 2);
          n_inst = 16;
        }
      }
      // fill g - DEPTH has landed when at most the instructions of the DEPTH younger fills are still out
      int younger = n_inst;
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d) younger += hist[d];
      wait_vm(younger);
      if (g >= DEPTH * NLOAD) { const int h = g - DEPTH * NLOAD; ready[h % RING] = h / RING + 1; }
#pragma unroll
      for (int d = DEPTH - 2; d > 0; --d) hist[d] = hist[d - 1];
      hist[0] = n_inst;
    }
    return;
  }

  // ---- CONSUMERS: wave c = 0..2 retires global slots g = c, c + 3, ...
  const int c = wave - NLOAD;
  u32x4_t xr[6] = {}, ar[16] = {};
  int have_x = 0, have_a = 0;                           // generation of the register copies
  for (int g = c; g < total; g += 3) {
    const int L = g / SLOTS_PER_LAYER, s = g % SLOTS_PER_LAYER, slot = g % RING, gen = g / RING + 1;
    const bool gu = s < GU_SLOTS;
    // the vector this slot multiplies: gathered once per layer by the first consumer that needs it
    if (!(p.mode & 1) && gu && have_x != L + 1) {
      if (lds_poll(&vec_gen[0]) != L + 1) {
        if (c == (L * SLOTS_PER_LAYER) % 3) {                                       // the consumer that owns the layer's first slot gathers
          if (p.stamps && L == p.stamp_layer + 1 && lane == 0) p.stamps[cu * 8 + 6] = wall_clock64();  // x-gather (next layer) begins
          gather<H / 256>(p.X[L & 1], xs, L + 1, lane);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          vec_gen[0] = L + 1;
          if (p.stamps && lane == 0 && L == p.stamp_layer) p.stamps[cu * 8 + 4] = wall_clock64();      // x of this layer in LDS
          if (p.stamps && lane == 0 && L == p.stamp_layer + 1) p.stamps[cu * 8 + 3] = wall_clock64();  // x of the next layer in LDS
        } else {
          while (lds_poll(&vec_gen[0]) != L + 1) __builtin_amdgcn_s_sleep(1);
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) xr[j] = ((const u32x4_t*)xs)[j * 64 + lane];
      have_x = L + 1;
    }
    if (!(p.mode & 1) && !gu && have_a != L + 1) {
      if (lds_poll(&vec_gen[1]) != L + 1) {
        if (c == (L * SLOTS_PER_LAYER + GU_SLOTS) % 3) {
          if (p.stamps && L == p.stamp_layer && lane == 0) p.stamps[cu * 8 + 5] = wall_clock64();      // a-gather begins
          gather<I / 256>(p.A[L & 1], as, L + 1, lane);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          vec_gen[1] = L + 1;
          if (p.stamps && L == p.stamp_layer && lane == 0) p.stamps[cu * 8 + 1] = wall_clock64();      // a in LDS
        } else {
          while (lds_poll(&vec_gen[1]) != L + 1) __builtin_amdgcn_s_sleep(1);
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) ar[j] = ((const u32x4_t*)as)[j * 64 + lane];
      have_a = L + 1;
    }
    while (lds_poll(&ready[slot]) != gen) __builtin_amdgcn_s_sleep(1);
    const u32x4_t* w = (const u32x4_t*)(ring + slot * SLOT_BYTES);
    if (p.mode & 2) { if (lane == 0) freed[slot] = gen; continue; }
    if (gu) {
      float g0 = 0.f, u0 = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        g0 = dot8(w[j * 64 + lane], xr[j], g0);
        u0 = dot8(w[384 + j * 64 + lane], xr[j], u0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) freed[slot] = gen;                                             // (the slot's bytes are in registers / consumed)
      g0 = wave_sum(g0); u0 = wave_sum(u0);
      if (lane == 0) {
        const float gt = bf16_round(g0), up = bf16_round(u0);
        st_gran(p.A[L & 1] + cu * GU_SLOTS + s, bf16_round(gt * bf16_round(1.f / (1.f + __expf(-gt)))) * up, L + 1);
        if (p.stamps && L == p.stamp_layer && s == GU_SLOTS - 1) p.stamps[cu * 8 + 0] = wall_clock64();   // last a granule of this CU out
      }
    } else {
      float d0 = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) d0 = dot8(w[j * 64 + lane], ar[j], d0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) freed[slot] = gen;
      d0 = wave_sum(d0);
      if (lane == 0) {
        const int row = cu * DN_ROWS + (s - GU_SLOTS);
        st_gran(p.X[(L + 1) & 1] + row, bf16_to_f32(xs[row]) + bf16_round(d0), L + 2);
        if (p.stamps && L == p.stamp_layer && s == SLOTS_PER_LAYER - 1) p.stamps[cu * 8 + 2] = wall_clock64();   // last x granule of this CU out
      }
    }
  }
}

static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t tobf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const int NL = argc > 1 ? atoi(argv[1]) : 32;
  EngP p;
  std::vector<uint16_t> hgu((size_t)2 * I * H), hd((size_t)H * I);
  srand(1);
  auto rnd = [] { return ((rand() & 0xffff) / 65536.f - 0.5f) * 0.04f; };
  for (auto& v : hgu) v = tobf(rnd());
  for (auto& v : hd) v = tobf(rnd() * 0.5f);
  for (int i = 0; i < NWSET; ++i) {
    bf16_t *a, *b;
    CK(hipMalloc(&a, hgu.size() * 2)); CK(hipMalloc(&b, hd.size() * 2));
    CK(hipMemcpy(a, hgu.data(), hgu.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hd.data(), hd.size() * 2, hipMemcpyHostToDevice));
    p.Wgu[i] = a; p.Wd[i] = b;
  }
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&p.A[i], I * 4)); CK(hipMalloc(&p.X[i], H * 4)); }
  std::vector<uint16_t> hx(H);
  for (auto& v : hx) v = tobf(((rand() & 0xffff) / 65536.f - 0.5f) * 2.f);
  std::vector<uint32_t> gx(H);
  CK(hipMalloc(&p.stamps, NCU * 8 * sizeof(long long)));
  p.stamp_layer = 10;
  CK(hipFuncSetAttribute((const void*)k_mlp_engine, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 8; ++rep) {
    p.n_layers = rep == 0 ? 2 : NL;
    p.mode = rep < 4 ? 0 : rep < 6 ? 1 : 3;
    for (int i = 0; i < 2; ++i) { CK(hipMemset(p.A[i], 0, I * 4)); CK(hipMemset(p.X[i], 0, H * 4)); }
    for (int i = 0; i < H; ++i) gx[i] = hx[i] | (1u << 16);
    CK(hipMemcpy(p.X[0], gx.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mlp_engine, dim3(NCU), dim3(64 * (NLOAD + 3)), LDS_BYTES, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 0) {
      // two layers against the host (same weights in every set): x2 = f(f(x0))
      std::vector<uint32_t> out(H);
      CK(hipMemcpy(out.data(), p.X[0], H * 4, hipMemcpyDeviceToHost));
      std::vector<float> x(H), a(I);
      for (int i = 0; i < H; ++i) x[i] = bf(hx[i]);
      for (int L = 0; L < 2; ++L) {
        for (int n = 0; n < I; ++n) {
          double g = 0, u = 0;
          for (int k = 0; k < H; ++k) { g += (double)bf(hgu[(size_t)n * H + k]) * x[k]; u += (double)bf(hgu[(size_t)(I + n) * H + k]) * x[k]; }
          const float gt = bf(tobf((float)g)), up = bf(tobf((float)u));
          a[n] = bf(tobf(bf(tobf(gt * bf(tobf(1.f / (1.f + expf(-gt)))))) * up));
        }
        std::vector<float> xn(H);
        for (int n = 0; n < H; ++n) {
          double d = 0;
          for (int k = 0; k < I; ++k) d += (double)bf(hd[(size_t)n * I + k]) * a[k];
          xn[n] = bf(tobf(x[n] + bf(tobf((float)d))));
        }
        x = xn;
      }
      double worst = 0, scale = 0;
      int bad_tag = 0;
      for (int i = 0; i < H; ++i) {
        if ((out[i] >> 16) != 3) ++bad_tag;
        worst = fmax(worst, fabs(bf(out[i] & 0xffff) - x[i])); scale = fmax(scale, fabs(x[i]));
      }
      printf("check (2 layers vs host): max |diff| %.4f of max |x| %.3f, %d granules with a wrong tag\n", worst, scale, bad_tag);
    } else {
      if (rep == 3) {
        std::vector<long long> st(NCU * 8);
        CK(hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        const char* names[7] = {"last a granule out", "a in LDS", "last x granule out", "next x in LDS", "x in LDS (layer start)", "a-gather begins", "x-gather begins"};
        long long t0 = st[4];
        for (int c = 0; c < NCU; ++c) t0 = std::min(t0, st[c * 8 + 4]);
        const int order[7] = {4, 0, 5, 1, 2, 6, 3};
        for (int k : order) {
          double mn = 1e30, mx = -1e30, sum = 0;
          for (int c = 0; c < NCU; ++c) { const double v = (st[c * 8 + k] - t0) / 100.0; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
          printf("  layer %d, %-24s min %6.2f  mean %6.2f  max %6.2f us after the first CU had its x\n", p.stamp_layer, names[k], mn, sum / NCU, mx);
        }
      }
      printf("engine (mode %d): %d layers in %.3f ms = %.2f us per MLP pair (151 MB: %.2f TB/s)\n", p.mode, NL, ms, ms * 1e3 / NL, 151.0 / (ms * 1e3 / NL));
    }
  }
  return 0;
}
