"""How many leading rows of a prompt-sized GEMM should take the 256 x 256 kernel?  The launcher's cost model (gemm_big_rows) against
pinned splits, per shape of the prefill, timed alternately.  Run on the GPU box: python tools/gemm_rows_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit

V, S = 17 * 577, 2531
CASES = [("vit qkv+bias", V, 3072, 1024, ops.EPI_BIAS, False), ("vit out+resid_f32", V, 1024, 1024, ops.EPI_BIAS_RESID_F32, True),
         ("vit fc1 qgelu", V, 4096, 1024, ops.EPI_BIAS_QGELU, False), ("vit fc2+resid_f32", V, 1024, 4096, ops.EPI_BIAS_RESID_F32, True),
         ("dec qkv", S, 9216, 3072, ops.EPI_NONE, False), ("dec o_proj+resid", S, 3072, 3072, ops.EPI_RESID_BF16, False),
         ("dec gate_up silu", S, 8192, 3072, ops.EPI_SILU_MUL, False), ("dec down+resid", S, 3072, 8192, ops.EPI_RESID_BF16, False)]
for name, M, N, K, epi, f32res in CASES:
    rows_w = 2 * N if epi == ops.EPI_SILU_MUL else N
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(rows_w, K, device="cuda").bfloat16() * 0.02 for _ in range(4)]
    bias = torch.randn(N, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32res else torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_QGELU, ops.EPI_BIAS_RESID_F32):
        kw["bias"] = bias
    if epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_RESID_BF16):
        kw.update(resid=res, out=res)
    mt = (M + 255) // 256
    splits = [("auto", -1), ("all small", 0)] + [(f"{r} big rows", r) for r in sorted({(mt - 2) * 256, (mt - 1) * 256, mt * 256}) if r > 0]
    t = {n: [] for n, _ in splits}
    for rep in range(3):
        for n, r in splits:
            old = ops.set_tuning("gemm_big_rows", r)
            t[n].append(timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), 4, iters=10))
            ops.set_tuning("gemm_big_rows", old)
    fl = 2.0 * M * N * K * (2 if epi == ops.EPI_SILU_MUL else 1)
    print(f"{name:20s} M={M} N={N} K={K}: " + "  ".join(f"{n}: {sorted(v)[1] * 1e3:6.1f} us ({fl / sorted(v)[1] / 1e9:5.0f} TF/s)" for n, v in t.items()), flush=True)
