"""256 x 256 GEMM: ping-pong K loop (p3v_gemm256pp.hip, gemm_pp=1) against the round-2..4 loop (gemm_pp=0), per prefill shape, alternated.
python tools/gemm_pp_probe.py [name filter]"""
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
         ("dec gate_up silu", S, 8192, 3072, ops.EPI_SILU_MUL, False), ("dec down+resid", S, 3072, 8192, ops.EPI_RESID_BF16, False),
         ("square 4096", 4096, 4096, 4096, ops.EPI_NONE, False), ("square 8192", 8192, 8192, 8192, ops.EPI_NONE, False),
         ("c3 qkv 32k rows", 32768, 9216, 3072, ops.EPI_NONE, False)]
only = sys.argv[1:]
for name, M, N, K, epi, f32res in CASES:
    if only and not any(o in name for o in only):
        continue
    rows_w = 2 * N if epi == ops.EPI_SILU_MUL else N
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(rows_w, K, device="cuda").bfloat16() * 0.02 for _ in range(3)]
    bias = torch.randn(N, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32res else torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_QGELU, ops.EPI_BIAS_GELU, ops.EPI_BIAS_RESID_F32):
        kw["bias"] = bias
    if epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_RESID_BF16):
        kw.update(resid=res, out=res)
    line = []
    for big in (-1, 1000000):
        t = {0: [], 1: []}
        outs = {}
        o_big = ops.set_tuning("gemm_big_rows", big)
        for rep in range(3):
            for pp in (0, 1):
                old = ops.set_tuning("gemm_pp", pp)
                if rep == 0:
                    res.zero_()
                    outs[pp] = ops.gemm(A, Ws[0], epi, **kw).float().clone()
                t[pp].append(timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), 3, iters=10))
                ops.set_tuning("gemm_pp", old)
        ops.set_tuning("gemm_big_rows", o_big)
        fl = 2.0 * M * N * K * (2 if epi == ops.EPI_SILU_MUL else 1)
        eq = bool((outs[0] == outs[1]).all())
        line.append(f"{'auto rows' if big < 0 else 'all big '}: loop {sorted(t[0])[1] * 1e3:6.1f} us ({fl / sorted(t[0])[1] / 1e9:5.0f}) ping-pong {sorted(t[1])[1] * 1e3:6.1f} us ({fl / sorted(t[1])[1] / 1e9:5.0f} TF/s) same bits {eq}")
    print(f"{name:20s} M={M} N={N} K={K}: " + " | ".join(line), flush=True)
