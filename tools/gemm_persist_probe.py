"""Per-tile vs persistent 256x256-tile GEMM, A/B interleaved (the chip's clock follows its power / thermal state: a variant
measured after another one is not comparable -- alternate them and take medians)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit
big = len(sys.argv) > 1 and sys.argv[1] == "big"
KNOB = sys.argv[2] if len(sys.argv) > 2 else "gemm_persistent"
if big:
    ops.set_tuning("gemm_big_rows", 1 << 20)
shapes = ((1792, 9216, 3072), (2048, 9216, 3072), (2560, 9216, 3072), (9728, 3072, 1024), (9728, 1024, 4096), (4096, 4096, 4096)) if big else \
         ((2531, 9216, 3072), (2531, 3072, 3072), (2531, 3072, 8192), (9809, 3072, 1024), (9809, 1024, 1024), (9809, 4096, 1024), (9809, 1024, 4096))
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(4)]
    t = {0: [], 1: []}
    for rep in range(6):
        for persist in (0, 1):
            ops.set_tuning(KNOB, persist)
            t[persist].append(timeit(lambda i: ops.gemm(A, Ws[i], ops.EPI_NONE), 4, iters=8))
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"M={M:5d} N={N:5d} K={K:5d}: {KNOB}=0 {a*1e3:7.1f} us ({2*M*N*K/a/1e9:6.0f} TF/s)   =1 {b*1e3:7.1f} us ({2*M*N*K/b/1e9:6.0f} TF/s)   ratio {b/a:.3f}", flush=True)
