"""Writes build/slot_bench.hip: the instruction mix of one 'slot' of an intra-wave-pipelined attention loop (26 MFMA 16x16x32 +
~55 VALU of which 16 v_exp and 8 v_cvt_pk + 12 ds_read_b128), hand-interleaved in inline asm, two waves per SIMD.  Upper bound
of what the hardware gives that structure; compare with 26 x 16 = 416 MFMA cycles per slot and wave."""
import sys
def slot(variant):
    lines = []
    valu = []
    # 23 'reduction' ops (dependent pairs of chains), 16 exps, 8 cvts -- on registers v[100..131] (s), results v[132..139]
    for i in range(11): valu.append(f"v_max3_f32 v{140 + (i & 3)}, v{100 + 3 * (i % 5)}, v{101 + 3 * (i % 5)}, v{140 + (i & 3)}")
    for i in range(12): valu.append(f"v_max_f32 v{144 + (i & 3)}, v{140 + (i & 3)}, v{144 + (i & 3)}")
    for i in range(16): valu.append(f"v_exp_f32 v{100 + i}, v{100 + i}")
    for i in range(8): valu.append(f"v_cvt_pk_bf16_f32 v{132 + i}, v{100 + 2 * i}, v{101 + 2 * i}")
    nv = len(valu)
    vi = 0
    for k in range(26):
        acc = 4 * (k % 13)                                   # 13 accumulators v[0..51]
        a = 60 + 4 * (k % 6)                                 # A fragments v[60..83] (the ds_read targets)
        lines.append(f"v_mfma_f32_16x16x32_bf16 v[{acc}:{acc + 3}], v[{a}:{a + 3}], v[88:91], v[{acc}:{acc + 3}]")
        if variant != "mfma":
            n = (nv * (k + 1)) // 26 - (nv * k) // 26
            for _ in range(n):
                lines.append(valu[vi]); vi += 1
        if variant not in ("mfma", "nolds") and k % 2 == 0 and k < 24:
            tgt = 60 + 4 * ((k // 2 + 3) % 6)
            lines.append(f"ds_read_b128 v[{tgt}:{tgt + 3}], v150 offset:{(k // 2) * 1024}")
        if variant == "salu":                                   # the compiled loop carries ~25 scalar instructions per slot
            lines.append(f"s_add_u32 s{20 + k % 4}, s{20 + k % 4}, 1")
        if variant == "waitcnt":                                # ... and a counted wait in front of every MFMA
            lines.append(f"s_waitcnt lgkmcnt({min(15, 6)})")
        if variant == "branch" and k % 8 == 4:                  # ... and 3 taken branches per slot
            lines += ["s_cmp_eq_u32 s20, s20", f"s_cbranch_scc1 .Lsb_{variant}_{k}%=", "s_nop 0", f".Lsb_{variant}_{k}%=:"]
        if variant == "barrier" and k == 13:                    # one s_barrier per slot
            lines.append("s_barrier")
    if variant not in ("mfma", "nolds"): lines.append("s_waitcnt lgkmcnt(0)")
    return lines
src = r'''#include <hip/hip_runtime.h>
#include <cstdio>
'''
for variant in ("mfma", "nolds", "full", "salu", "waitcnt", "branch", "barrier"):
    body = "\\n\t".join(slot(variant))
    clob = ", ".join([f'"v{i}"' for i in range(0, 152)] + ['"s20"', '"s21"', '"s22"', '"s23"', '"scc"'])
    src += f'''
__global__ __launch_bounds__(512, 1) void k_{variant}(float* out, int iters) {{
  __shared__ float lds[16384];
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 8192] = 1.f;
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 16;
  asm volatile("v_mov_b32 v150, %0" :: "v"(addr) : "v150");
  for (int it = 0; it < iters; ++it) {{
    asm volatile("{body}" ::: {clob}, "memory");
  }}
  float r;
  asm volatile("v_mov_b32 %0, v0" : "=v"(r));
  if (r == 123.456f) out[threadIdx.x] = r + lds[threadIdx.x ^ 1];
}}
'''
src += r'''
template <typename K> static void run(const char* name, K kern, float* out, int threads) {
  const int iters = 4000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  kern<<<256, threads>>>(out, iters);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(a);
    kern<<<256, threads>>>(out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("%-40s %d waves/SIMD: %7.1f ns per slot and wave (26 MFMAs: 26 x 16 cycles = %.0f ns at 2.1 GHz)\n", name, threads / 256, best * 1e6 / iters, 26 * 16 / 2.1);
}
int main() {
  float* out;
  (void)hipMalloc(&out, 4096);
  for (int t : {256, 512}) {
    run("26 MFMA", k_mfma, out, t);
    run("26 MFMA + 47 VALU", k_nolds, out, t);
    run("26 MFMA + 47 VALU + 12 ds_read_b128", k_full, out, t);
    run("  + 26 SALU", k_salu, out, t);
    run("  + s_waitcnt lgkmcnt before every MFMA", k_waitcnt, out, t);
    run("  + 3 taken branches", k_branch, out, t);
    run("  + 1 s_barrier", k_barrier, out, t);
  }
  return 0;
}
'''
open(sys.argv[1], "w").write(src)
