"""dma / dma (pre-scaled) / interleaved kernel at the ViT's shape and at a few batched decoder shapes."""
import sys, statistics, torch
sys.path.insert(0, "/root/repo")
from phi_3_vision_mlx_amd import ops
def t(B, L, nh, hd, causal, pp, il, pre):
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE if pre else 1.0)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    ops.set_tuning("attn_pp", pp); ops.set_tuning("attn_il", il)
    f = lambda: ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, causal, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
    ts = []
    for r in range(5):
        f(); torch.cuda.synchronize()
        a, b = ops.Event(), ops.Event(); a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_ms(b) / 20 * 1e3)
    return statistics.median(ts)
for (B, L, nh, hd, c) in ((17, 577, 16, 64, False), (8, 512, 32, 96, True), (4, 2531, 32, 96, True), (8, 1024, 32, 96, True)):
    print(f"B={B} L={L} nh={nh} hd={hd} causal={c}: dma {t(B,L,nh,hd,c,0,0,False):7.1f} us   dma(prescaled) {t(B,L,nh,hd,c,0,0,True):7.1f}   il {t(B,L,nh,hd,c,1,1,True):7.1f} us", flush=True)
