// Feasibility probe (round 4): a chain of dependent weight-streaming GEMVs as ONE persistent kernel.  Every workgroup stays
// resident for all stages; between stages a grid-wide barrier, and -- the point -- each wave requests its NEXT stage's weight
// rows BEFORE it waits at the barrier, so HBM keeps streaming while the hand-over (write-through stores, barrier, uncached
// re-read of the vector) is in flight.  Compared with the same stages as one graph of plain dependent launches.
// Stage = y = tanh(W x), W 4096 x 3072 bf16 (25.2 MB), the next stage reads the first 3072 outputs; 512 workgroups x 4 waves, 2 rows per wave.
//   hipcc --offload-arch=gfx950 -O3 -o persistent_chain persistent_chain.hip && ./persistent_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 3072, NR = 4096, NS = 160, NWB = 24, TPB = 256, NBLK = 512, NCW = TPB / 64, CH = D / 8 / 64;   // K = D, NR rows; 6 chunks per lane and row
constexpr int XS = NR / 2;                                      // words between the vectors of consecutive stages (a stage reads the first D / 2)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float lo(unsigned x) { return __builtin_bit_cast(float, x << 16); }
__device__ __forceinline__ float hi(unsigned x) { return __builtin_bit_cast(float, x & 0xffff0000u); }
__device__ __forceinline__ unsigned pack(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
struct Ptrs { const u32x4* W[NWB]; };

__device__ __forceinline__ void load_rows(u32x4 (&w)[2][CH], const u32x4* W, int row0, int lane) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < CH; ++c) w[r][c] = __builtin_nontemporal_load(W + (size_t)(row0 + r) * (D / 8) + c * 64 + lane);
}
__device__ __forceinline__ unsigned dot_rows(const u32x4 (&w)[2][CH], const u32x4* xs, int lane) {
  float acc[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const u32x4 x = xs[c * 64 + lane];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[r] += lo(w[r][c][j]) * lo(x[j]) + hi(w[r][c][j]) * hi(x[j]);
  }
  return pack(tanhf(wave_sum(acc[0])), tanhf(wave_sum(acc[1])));
}

// one launch per stage
__global__ void __launch_bounds__(TPB) k_stage(const u32x4* __restrict__ W, const unsigned* xin, unsigned* xout) {
  __shared__ u32x4 xs[D / 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = (blockIdx.x * NCW + wave) * 2;
  u32x4 w[2][CH];
  load_rows(w, W, row0, lane);
  for (int i = tid; i < D / 8; i += TPB) xs[i] = *(const u32x4*)(xin + i * 4);
  __syncthreads();
  const unsigned o = dot_rows(w, xs, lane);
  if (lane == 0) xout[row0 >> 1] = o;
}

// MODE 0: barrier = one atomic counter per stage.  MODE 1: two-level flags (no atomics): every workgroup stores its own flag
// word, 16 group leaders watch 32 flags each and store a group word, every workgroup watches the 16 group words.
// Waves 0..3 stream and reduce; wave 4 is the COMMUNICATION wave: it alone polls, re-reads the vector and publishes the
// workgroup's outputs + flag, so none of that queues behind weight loads (memory returns a wave's loads in order) and the
// "stores acknowledged" wait before the flag does not drain the computing waves' prefetch.
template <int MODE, bool PREFETCH>
__global__ void __launch_bounds__(TPB + 64, 4) k_chain(const u32x4* const* __restrict__ PW, unsigned* x, unsigned* sync, int* err, long long* stamp) {
  __shared__ u32x4 xs[D / 8];
  __shared__ unsigned outw[NCW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bx = blockIdx.x;
  const int row0 = (bx * NCW + (wave % NCW)) * 2;
  constexpr int XW = D / 2;                                          // words per vector
  u32x4 w[2][CH];
  bool fail = false;
  if (wave < NCW) load_rows(w, PW[0], row0, lane);
  for (int s = 0; s < NS; ++s) {
    const bool st = stamp && (bx == 0 || bx == NBLK - 1);
    long long* sp = stamp + s * 8 + (bx ? 4 : 0);
    if (wave == NCW) {
      if (st && lane == 0) sp[0] = wall_clock64();
      // ---- wait until stage s-1 is complete everywhere (stage 0: the host wrote x)
      if (s > 0 && !fail) {                                         // (a timeout is sticky: no more waiting after the first)
        int tries = 0;
        if (MODE == 0) {
          while (__hip_atomic_load(sync + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NBLK) {
            if (++tries > 20000) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        } else {
          unsigned* flags = sync + (size_t)(s - 1) * (NBLK + 64);
          if ((bx & 31) == 0) {                                       // group leader: my 32 workgroups' flags
            while (true) {
              const unsigned f = lane < 32 ? __hip_atomic_load(flags + bx + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
              if (!__any(f == 0u)) break;
              if (++tries > 20000) { fail = true; break; }
              __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(flags + NBLK + (bx >> 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          while (true) {
            const unsigned f = lane < NBLK / 32 ? __hip_atomic_load(flags + NBLK + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            if (!__any(f == 0u)) break;
            if (++tries > 40000) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (st && lane == 0) sp[1] = wall_clock64();
      // ---- the vector, past the caches, into LDS
      unsigned* xw = (unsigned*)xs;
      const unsigned* src = x + (size_t)s * XS;
#pragma unroll 1
      for (int j0 = 0; j0 < XW / 64; j0 += 4) {
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __hip_atomic_load(src + (j0 + j) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 4; ++j) xw[(j0 + j) * 64 + lane] = v[j];
      }
      if (st && lane == 0) sp[2] = wall_clock64();
    }
    __syncthreads();                                                  // A: x is in LDS
    if (wave < NCW) {
      if (!PREFETCH && s > 0) load_rows(w, PW[s % NWB], row0, lane);
      const unsigned o = dot_rows(w, xs, lane);
      if (lane == 0) outw[wave] = o;
      __builtin_amdgcn_sched_barrier(0);                             // (or the next stage's loads are hoisted above the reduction: 2 x 64 registers)
      // next stage's rows: requested now, they travel while the barrier resolves
      if (PREFETCH && s + 1 < NS) load_rows(w, PW[(s + 1) % NWB], row0, lane);
    }
    __syncthreads();                                                  // B: outputs are in LDS, xs is free again
    if (wave == NCW) {
      if (lane < NCW) __hip_atomic_store(x + (size_t)(s + 1) * XS + bx * NCW + lane, outw[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // write-through stores acknowledged
      if (lane == 0 && s + 1 < NS) {
        if (MODE == 0) __hip_atomic_fetch_add(sync + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(sync + (size_t)s * (NBLK + 64) + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (st && lane == 0) sp[3] = wall_clock64();
    }
  }
  if (wave == NCW && __any(fail) && lane == 0) atomicExch(err, 1);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  std::vector<unsigned short> hw((size_t)NR * D);
  srand(1);
  for (auto& v : hw) {
    const float f = ((rand() & 0xffff) / 65536.f - 0.5f) * 0.06f;
    unsigned u;
    memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  Ptrs P;
  for (int i = 0; i < NWB; ++i) {
    u32x4* w;
    CK(hipMalloc(&w, (size_t)NR * D * 2));
    for (size_t k = 0; k < hw.size(); k += 977) hw[k] ^= (unsigned short)(i + 1);
    CK(hipMemcpy(w, hw.data(), (size_t)NR * D * 2, hipMemcpyHostToDevice));
    P.W[i] = w;
  }
  const u32x4** dW;
  CK(hipMalloc(&dW, sizeof(P)));
  CK(hipMemcpy(dW, &P, sizeof(P), hipMemcpyHostToDevice));
  unsigned *x, *sync;
  int* err;
  long long* stamps;
  const size_t xw = XS, sync_words = (size_t)NS * (NBLK + 64);
  CK(hipMalloc(&x, (NS + 1) * xw * 4));
  CK(hipMalloc(&sync, sync_words * 4));
  CK(hipMalloc(&err, 4));
  CK(hipMemset(err, 0, 4));
  CK(hipMalloc(&stamps, NS * 8 * 8));
  std::vector<unsigned> hx(D / 2, 0x3c003c00u);
  hipStream_t s1;
  CK(hipStreamCreate(&s1));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_chain<1, true>, TPB + 64, 0));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  printf("k_chain: %d workgroups resident per CU x %d CUs (grid %d)\n", occ, pr.multiProcessorCount, NBLK);
  if (occ * pr.multiProcessorCount < NBLK + NBLK / 4) { printf("not enough head-room for the grid to be co-resident\n"); return 1; }
  auto capture = [&](int mode) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    if (mode == 0) {
      for (int i = 0; i < NS; ++i) hipLaunchKernelGGL(k_stage, dim3(NBLK), dim3(TPB), 0, s1, P.W[i % NWB], x + i * xw, x + (i + 1) * xw);
    } else {
      CK(hipMemsetAsync(sync, 0, sync_words * 4, s1));
      if (mode == 1) hipLaunchKernelGGL((k_chain<0, true>), dim3(NBLK), dim3(TPB + 64), 0, s1, dW, x, sync, err, stamps);
      if (mode == 2) hipLaunchKernelGGL((k_chain<1, true>), dim3(NBLK), dim3(TPB + 64), 0, s1, dW, x, sync, err, stamps);
      if (mode == 3) hipLaunchKernelGGL((k_chain<1, false>), dim3(NBLK), dim3(TPB + 64), 0, s1, dW, x, sync, err, stamps);
    }
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
  };
  hipGraphExec_t gx[4];
  for (int m = 0; m < 4; ++m) gx[m] = capture(m);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned> ref(xw), got(xw);
  auto run = [&](const char* name, int mode, std::vector<unsigned>& out) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemsetAsync(x, 0, (NS + 1) * xw * 4, s1));
      CK(hipMemcpyAsync(x, hx.data(), D / 2 * 4, hipMemcpyHostToDevice, s1));
      CK(hipMemsetAsync(stamps, 0, NS * 8 * 8, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      CK(hipGraphLaunch(gx[mode], s1));
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    CK(hipMemcpy(out.data(), x + NS * xw, xw * 4, hipMemcpyDeviceToHost));
    int herr;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemset(err, 0, 4));
    printf("%-58s %8.3f ms for %d stages = %6.2f us per stage (%5.2f TB/s)%s\n", name, best, NS, best * 1e3 / NS,
           (double)NR * D * 2 * NS / (best * 1e-3) / 1e12, herr ? "  [BARRIER TIMEOUT]" : "");
    if (mode > 0) {
      std::vector<long long> hs(NS * 8);
      CK(hipMemcpy(hs.data(), stamps, NS * 8 * 8, hipMemcpyDeviceToHost));
      printf("  stages 40..43, 100 MHz ticks from stage 40 (first / last workgroup: stage entry, barrier passed, x in LDS, arrived):\n");
      for (int i = 40; i < 44; ++i)
        printf("    stage %d: wg0 %5lld %5lld %5lld %5lld   wg%d %5lld %5lld %5lld %5lld\n", i, hs[i * 8] - hs[320], hs[i * 8 + 1] - hs[320], hs[i * 8 + 2] - hs[320],
               hs[i * 8 + 3] - hs[320], NBLK - 1, hs[i * 8 + 4] - hs[320], hs[i * 8 + 5] - hs[320], hs[i * 8 + 6] - hs[320], hs[i * 8 + 7] - hs[320]);
      printf("  same result as plain launches: %s\n", memcmp(ref.data(), out.data(), xw * 4) == 0 ? "yes" : "NO");
    }
  };
  run("one graph, plain dependent launches", 0, ref);
  run("persistent, atomic-counter barrier, prefetch", 1, got);
  run("persistent, two-level flag barrier, prefetch", 2, got);
  run("persistent, two-level flag barrier, NO prefetch", 3, got);
  run("one graph, plain dependent launches (again)", 0, ref);
  return 0;
}
