"""Per-tile vs persistent 256x256-tile GEMM on whole-tile shapes (all rows on the big-tile kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit
ops.set_tuning("gemm_big_rows", 1 << 20)
for M, N, K in ((1792, 9216, 3072), (2048, 9216, 3072), (2560, 9216, 3072), (2304, 3072, 3072), (2560, 3072, 8192), (9728, 3072, 1024), (9728, 1024, 1024), (9728, 1024, 4096), (2048, 16384, 3072)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(4)]
    out = []
    for persist in (0, 1):
        ops.set_tuning("gemm_persistent", persist)
        ms = timeit(lambda i: ops.gemm(A, Ws[i], ops.EPI_NONE), 4)
        out.append(ms)
    tiles = (M // 256) * (N // 256)
    print(f"M={M:5d} N={N:5d} K={K:5d} tiles {tiles:4d}: per-tile {out[0]*1e3:7.1f} us ({2*M*N*K/out[0]/1e9:6.0f} TF/s)   persistent {out[1]*1e3:7.1f} us ({2*M*N*K/out[1]/1e9:6.0f} TF/s)", flush=True)
