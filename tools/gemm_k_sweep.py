"""How much of a prompt-sized GEMM is its K loop and how much is per-tile fixed cost (cold fetch, epilogue, launch)?  Time against K
at fixed M, N: the intercept of the fit is the fixed part.  python tools/gemm_k_sweep.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit

V, S = 17 * 577, 2531
CASES = [("vit qkv+bias", V, 3072, ops.EPI_BIAS, False), ("vit out/fc2+resid_f32", V, 1024, ops.EPI_BIAS_RESID_F32, True),
         ("vit fc1 qgelu", V, 4096, ops.EPI_BIAS_QGELU, False),
         ("dec qkv", S, 9216, ops.EPI_NONE, False), ("dec o/down+resid", S, 3072, ops.EPI_RESID_BF16, False),
         ("dec gate_up silu", S, 8192, ops.EPI_SILU_MUL, False)]
KS = [64, 256, 512, 1024, 2048, 3072, 4096]
for name, M, N, epi, f32res in CASES:
    rows_w = 2 * N if epi == ops.EPI_SILU_MUL else N
    bias = torch.randn(N, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32res else torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_QGELU, ops.EPI_BIAS_RESID_F32):
        kw["bias"] = bias
    if epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_RESID_BF16):
        kw.update(resid=res, out=res)
    ts = []
    for K in KS:
        A = torch.randn(M, K, device="cuda").bfloat16()
        Ws = [torch.randn(rows_w, K, device="cuda").bfloat16() * 0.02 for _ in range(4)]
        ts.append(sorted(timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), 4, iters=10) for _ in range(3))[1] * 1e3)
    b, a = np.polyfit(KS[2:], ts[2:], 1)
    out_mb = M * N * (4 if f32res else 2) * (2 if epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_RESID_BF16) else 1) / 1e6
    print(f"{name:24s} M={M} N={N}: " + " ".join(f"K={k}:{t:6.1f}" for k, t in zip(KS, ts)) +
          f" us | fit (K>=512): {a:5.1f} us + {b * 1024:5.1f} us per 1024 k ({2 * M * N * (2 if epi == ops.EPI_SILU_MUL else 1) * 1024 / b / 1024 / 1e9:5.0f} TF/s slope); epilogue moves {out_mb:.0f} MB", flush=True)
