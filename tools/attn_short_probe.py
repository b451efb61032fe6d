"""dma / ping-pong / interleaved (8 waves) kernel over prompt lengths 512 .. 4096, B = 1, pinned and alternated."""
import sys, os, statistics, torch
sys.path.insert(0, "/root/repo")
from phi_3_vision_mlx_amd import ops
def t(L, pp, il, nh=32, hd=96):
    q = (torch.randn(1, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = torch.randn(1, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(1, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(1, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    ops.set_tuning("attn_pp", pp); ops.set_tuning("attn_il", il)
    f = lambda: ops.attention(q, out, 1, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=True)
    ts = []
    for r in range(5):
        f(); torch.cuda.synchronize()
        a, b = ops.Event(), ops.Event(); a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_ms(b) / 20 * 1e3)
    return statistics.median(ts)
for L in (512, 768, 1024, 1280, 1536, 1792, 2048, 2531, 3072, 4096):
    print(f"L={L}: dma {t(L,0,0):7.1f} us   pp {t(L,1,0):7.1f} us   il {t(L,1,1):7.1f} us", flush=True)
