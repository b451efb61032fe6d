"""Short-prompt (M = 17..256) projections: the split-K launch policy knobs against each other.  python tools/splitk_probe.py [M]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CASES = [("qkv", 9216, 3072, ops.EPI_NONE), ("o_proj", 3072, 3072, ops.EPI_RESID_BF16), ("gate_up", 8192, 3072, ops.EPI_SILU_MUL),
         ("down", 3072, 8192, ops.EPI_RESID_BF16)]
SETTINGS = [(256, 8), (512, 8), (512, 16), (1024, 16), (1024, 32), (2048, 32)]
tot = {s: 0.0 for s in SETTINGS}
for name, N, K, epi in CASES:
    rows_w = 2 * N if epi == ops.EPI_SILU_MUL else N
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(rows_w, K, device="cuda").bfloat16() * 0.02 for _ in range(6)]
    res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=res, out=res) if epi == ops.EPI_RESID_BF16 else {}
    line = []
    for wgs, ms in SETTINGS:
        o1, o2 = ops.set_tuning("gemm_splitk_wgs", wgs), ops.set_tuning("gemm_splitk_max_s", ms)
        t = sorted(timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), 6, iters=12) for _ in range(3))[1]
        ops.set_tuning("gemm_splitk_wgs", o1), ops.set_tuning("gemm_splitk_max_s", o2)
        tot[(wgs, ms)] += t
        line.append(f"wgs {wgs} S<={ms}: {t * 1e3:5.1f} us ({rows_w * K * 2 / t / 1e9:4.2f} TB/s)")
    print(f"M={M} {name:8s} N={N} K={K}: " + "  ".join(line), flush=True)
print("sum per layer: " + "  ".join(f"{s}: {t * 1e3:.1f} us" for s, t in tot.items()))
