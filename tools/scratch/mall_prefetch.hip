// Feasibility probe: can a persistent one-wave-per-CU streamer, running CONCURRENTLY with the decode step's kernel chain
// (second branch of the same hipGraph), pull the weights into the 256 MiB Infinity Cache just ahead of the kernels that
// read them, so that those kernels start warm (mall_probe: a warm 100 MB read is ~3.5 us shorter than a cold one)?
//   chain   : per layer 4 streaming kernels (56.6 / 18.9 / 100.7 / 50.3 MB, nt dwordx4, 2048 waves) + a 12 us
//             latency-bound stand-in for attention; each kernel publishes its index (one relaxed store)
//   streamer: 256 single-wave workgroups; lane l touches one dword per 64-byte sector (4 KiB per instruction, the
//             destination register is dead, so 32 instructions = 128 KiB are in flight per wave without registers);
//             it never runs more than LEAD bytes ahead of the published index
// hipcc --offload-arch=gfx950 -O3 -o mall_prefetch mall_prefetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_stream(const u32x4* __restrict__ p, size_t n16, int* progress, int tag, unsigned* sink) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const size_t per_wave = (n16 + gridDim.x * 4 - 1) / (gridDim.x * 4);          // contiguous run per wave, 16-byte units
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t b = w * per_wave, e = b + per_wave < n16 ? b + per_wave : n16;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = b + lane; i < e; i += 64 * 12) {
    u32x4 v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) v[u] = i + 64 * u < e ? __builtin_nontemporal_load(p + i + 64 * u) : (u32x4){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 12; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
__global__ void __launch_bounds__(256) k_fake_attn(int* progress, int tag, long long cycles) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
struct Seg { const char* ptr; size_t bytes; size_t cum; int tag; };   // cum = bytes of all earlier segments
template <bool FULL>
__global__ void __launch_bounds__(64) k_prefetch(const Seg* segs, int n_seg, const int* progress, const size_t* cum_by_tag,
                                                 size_t lead, unsigned* sink) {
  const int lane = threadIdx.x;
  constexpr size_t CHUNK = FULL ? (32 << 10) : (128 << 10);                       // per wave and iteration: 32 instructions x 4 KiB
  size_t chunk_id = blockIdx.x;                             // global chunk index over the concatenated segments
  size_t seg_first_chunk = 0;
  unsigned acc = 0;
  for (int s = 0; s < n_seg; ++s) {
    const Seg sg = segs[s];
    const size_t n_chunk = (sg.bytes + CHUNK - 1) / CHUNK;
    for (; chunk_id < seg_first_chunk + n_chunk; chunk_id += gridDim.x) {
      const size_t off = (chunk_id - seg_first_chunk) * CHUNK;
      // pacing: stay within `lead` bytes of what the chain has started to consume
      for (int spin = 0; spin < (1 << 22); ++spin) {
        const int tag = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sg.cum + off < cum_by_tag[tag] + lead) break;
        __builtin_amdgcn_s_sleep(64);
      }
      // 32 sector touches (lane l: byte 64*l of every 4-KiB slab) into ONE dead register, and the wait inside the same asm:
      // a load's destination may not be handed back to the compiler while the load is in flight
      const unsigned long long bv = (unsigned long long)(sg.ptr + off);                  // wave-uniform: force it into SGPRs
      const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bv >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)bv);
      unsigned voff = lane * (FULL ? 16 : 64), d;
      if (FULL) {
        u32x4 d4;
        asm volatile(
            "s_nop 4\n"
            ".rept 32\n"
            "global_load_dwordx4 %0, %1, %2\n"
            "v_add_u32 %1, 0x400, %1\n"
            ".endr\n"
            "s_waitcnt vmcnt(0)"
            : "=&v"(d4), "+v"(voff) : "s"(base) : "memory");
        d = d4[0];
      } else
      asm volatile(
          "s_nop 4\n"                     // VALU (v_readfirstlane) wrote the SGPR base: 5 wait states before a VMEM reads it
          ".rept 32\n"
          "global_load_dword %0, %1, %2\n"
          "v_add_u32 %1, 0x1000, %1\n"
          ".endr\n"
          "s_waitcnt vmcnt(0)"
          : "=&v"(d), "+v"(voff) : "s"(base) : "memory");
      acc ^= d;
    }
    seg_first_chunk += n_chunk;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int NL = 32;
  const size_t sizes[4] = {(size_t)9216 * 3072 * 2, (size_t)3072 * 3072 * 2, (size_t)16384 * 3072 * 2, (size_t)3072 * 8192 * 2};
  const size_t lead = (size_t)(argc > 1 ? atoi(argv[1]) : 128) << 20;
  const int n_pf_blocks = argc > 2 ? atoi(argv[2]) : 256;
  const bool full = argc > 3 && atoi(argv[3]);
  size_t total = 0;
  for (int l = 0; l < NL; ++l) for (int k = 0; k < 4; ++k) total += sizes[k];
  char* W; CK(hipMalloc(&W, total)); CK(hipMemset(W, 1, total));
  printf("W %p .. %p\n", W, W + total);
  int* progress; unsigned* sink; CK(hipMalloc(&progress, 256)); CK(hipMalloc(&sink, 4)); CK(hipMemset(progress, 0, 256));
  printf("progress %p sink %p\n", progress, sink);
  // order of the chain per layer: qkv(0) attn o_proj(1) gate_up(2) down(3); tags count kernels
  std::vector<Seg> segs; std::vector<size_t> cum_by_tag;
  size_t cum = 0; int tag = 0; size_t off = 0;
  struct K { int kind; const char* ptr; size_t bytes; int tag; };
  std::vector<K> chain;
  for (int l = 0; l < NL; ++l) {
    for (int k = 0; k < 4; ++k) {
      if (k == 1) { chain.push_back({1, nullptr, 0, tag}); cum_by_tag.push_back(cum); ++tag; }   // attention before o_proj
      segs.push_back({W + off, sizes[k], cum, tag});
      chain.push_back({0, W + off, sizes[k], tag});
      cum_by_tag.push_back(cum); ++tag;
      cum += sizes[k]; off += sizes[k];
    }
  }
  cum_by_tag.push_back(cum);
  Seg* d_segs; size_t* d_cum;
  CK(hipMalloc(&d_segs, segs.size() * sizeof(Seg))); CK(hipMemcpy(d_segs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_cum, cum_by_tag.size() * 8)); CK(hipMemcpy(d_cum, cum_by_tag.data(), cum_by_tag.size() * 8, hipMemcpyHostToDevice));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  hipEvent_t fork, join, a, b; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join)); CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // the chain is a graph on stream s; the streamer is launched eagerly on stream s2 next to every replay (two branches of ONE
  // hipGraph are executed one after the other by this runtime: 1.70 + 1.41 ms, and a paced streamer then waits for ever)
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(progress, 0, 4, s));
  for (auto& k : chain) {
    if (k.kind == 0) hipLaunchKernelGGL(k_stream, dim3(512), dim3(256), 0, s, (const u32x4*)k.ptr, k.bytes / 16, progress, k.tag, sink);
    else hipLaunchKernelGGL(k_fake_attn, dim3(1312), dim3(256), 0, s, progress, k.tag, (long long)1200);   // 100 MHz clock: 12 us
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int mode = 0; mode < 2; ++mode) {      // 0: chain only, 1: chain + streamer
    CK(hipGraphLaunch(ge, s)); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(progress, 0, 4)); CK(hipDeviceSynchronize());
      if (mode == 1 && full) hipLaunchKernelGGL(k_prefetch<true>, dim3(n_pf_blocks), dim3(64), 0, s2, d_segs, (int)segs.size(), progress, d_cum, lead, sink);
      else if (mode == 1) hipLaunchKernelGGL(k_prefetch<false>, dim3(n_pf_blocks), dim3(64), 0, s2, d_segs, (int)segs.size(), progress, d_cum, lead, sink);
      CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%s: %.3f ms per step (%.2f GB weights, lead %zu MB, %d streamer waves)\n", mode ? "chain + streamer" : "chain only      ",
           best, total / 1e9, lead >> 20, n_pf_blocks);
  }
  return 0;
}
