"""The 128 x 64-tile weight-streaming GEMM (p3v_gemm_skinny.hip) against the split-K path of the 128 x 128 kernel on the decoder's
projections at 17 .. 256 rows: values against a float64 product of the same bf16 operands, and time per launch over rotating weights
(4 x the matrix, so every launch streams from HBM).  Run on the GPU box: python tools/skinny_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops


def timeit(fn, n_rot, iters=20):
    """ms per call, the calls captured into one graph (eager ctypes launches would time the host)"""
    for i in range(n_rot): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i % n_rot)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

CASES = [("qkv", 9216, 3072, ops.EPI_NONE), ("o_proj+resid", 3072, 3072, ops.EPI_RESID_BF16),
         ("gate_up silu", 8192, 3072, ops.EPI_SILU_MUL), ("down+resid", 3072, 8192, ops.EPI_RESID_BF16)]
Ms = [int(m) for m in sys.argv[1:]] or [128, 17, 64, 100, 200, 256]
torch.manual_seed(0)
for M in Ms:
    tot = {"skinny": 0.0, "split-K 128x128": 0.0}
    for name, N, K, epi in CASES:
        rows_w = 2 * N if epi == ops.EPI_SILU_MUL else N
        A = torch.randn(M, K, device="cuda").bfloat16()
        Ws = [(torch.randn(rows_w, K, device="cuda") * 0.02).bfloat16() for _ in range(4)]
        res = torch.randn(M, N, device="cuda").bfloat16()
        kw = {"resid": res} if epi == ops.EPI_RESID_BF16 else {}
        outs, t = {}, {}
        variants = [("skinny", 0, 0), ("split-K 128x128", 1, 0)] + ([("skinny, 128-row tiles", 0, 1)] if M <= 64 else [])
        for label, off, tm128 in variants:
            old, old2 = ops.set_tuning("gemm_no_skinny", off), ops.set_tuning("gemm_skinny_tm128", tm128)
            outs[label] = ops.gemm(A, Ws[0], epi, **kw).float()
            t[label] = sorted(timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), 4, iters=20) for _ in range(3))[1]
            ops.set_tuning("gemm_no_skinny", old), ops.set_tuning("gemm_skinny_tm128", old2)
            tot[label] = tot.get(label, 0.0) + t[label]
        z = A.double() @ Ws[0].double().t()
        if epi == ops.EPI_SILU_MUL:
            g, u = z[:, :N].bfloat16().double(), z[:, N:].bfloat16().double()
            ref = (g * torch.sigmoid(g)).bfloat16().double() * u
        elif epi == ops.EPI_RESID_BF16:
            ref = res.double() + z.bfloat16().double()
        else:
            ref = z
        scale = ref.abs().max().item()
        err = {k: ((v.double() - ref).abs().max().item() / scale) for k, v in outs.items()}
        same = torch.equal(outs["skinny"], outs["split-K 128x128"])
        mb = rows_w * K * 2 / 1e6
        print(f"M={M:4d} {name:13s} N={N} K={K}: " + "  ".join(f"{k}: {v * 1e3:6.1f} us ({mb / v / 1e3:4.2f} TB/s) err {err[k]:.1e}" for k, v in t.items())
              + f"  bit-identical {same}", flush=True)
    print(f"M={M:4d} layer sum: " + "  ".join(f"{k}: {v * 1e3:6.1f} us" for k, v in tot.items()), flush=True)
