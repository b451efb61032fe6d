"""The table behind the attention launcher's rules: 128-query kernel / interleaved kernel with 4 waves / with 8 waves, pre-scaled
causal prompts, alternated in one process.  argv: B:L pairs (default: a sweep over B = 1, 2, 8)."""
import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from phi_3_vision_mlx_amd import ops
def t(B, L, mode, nh=32, hd=96):
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=True)
    res = {}
    ts = {"dma": [], "il4": [], "il8": []}
    for r in range(5):
        for m in ts:
            ops.set_tuning("attn_pp", 0 if m == "dma" else 1); ops.set_tuning("attn_il", 0 if m == "dma" else 1); ops.set_tuning("attn_il_waves", 4 if m == "il4" else 8)
            f(); torch.cuda.synchronize()
            a, b = ops.Event(), ops.Event(); a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize(); ts[m].append(a.elapsed_ms(b) / 20 * 1e3)
    return {m: statistics.median(v) for m, v in ts.items()}
import sys
CASES = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(B, L) for B in (1, 2, 8) for L in (128, 256, 384, 512, 768, 1024, 1280, 2531, 3072, 4096, 5120) if not (B == 8 and L > 2531)]
for B, L in CASES:
    if True:
        r = t(B, L, 0)
        print(f"B={B} L={L:5d}: dma {r['dma']:7.1f}   il4 {r['il4']:7.1f}   il8 {r['il8']:7.1f} us", flush=True)
