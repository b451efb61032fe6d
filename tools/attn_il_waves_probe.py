"""k_attn_prefill_il with 8 waves (256 queries) against 4 waves (128 queries, two workgroups per CU) per workgroup, alternated
in one process: where does the better balance of the small blocks beat their doubled DMA issue?  argv: B (default 1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops.set_tuning("attn_pp", 1), ops.set_tuning("attn_il", 1)
def t(L, nw, nh=32, hd=96):
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=True)
    ts = {4: [], 8: []}
    for r in range(5):
        for w in (8, 4):
            ops.set_tuning("attn_il_waves", w)
            f(); torch.cuda.synchronize()
            a, b = ops.Event(), ops.Event(); a.record()
            n = 20 if L <= 8192 else 4
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize(); ts[w].append(a.elapsed_ms(b) / n * 1e3)
    return statistics.median(ts[8]), statistics.median(ts[4])
for L in (1024, 1536, 1792, 2048, 2531, 3072, 4096, 6144, 8192, 16384):
    a, b = t(L, 0)
    print(f"B={B} L={L:6d}: 8 waves {a:8.1f} us   4 waves {b:8.1f} us   ({b / a:.3f})  256-query workgroups {B * 32 * ((L + 255) // 256)}", flush=True)
