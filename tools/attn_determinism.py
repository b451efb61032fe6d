"""Five repeats of every prompt-sized attention kernel on the same inputs at five shapes: number of output words that differ from
the first launch (must be 0; the test suite's version: test_attention_prefill_kernels_are_deterministic)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from phi_3_vision_mlx_amd import ops
def run(B, L, nh, hd, causal, pre, knobs, reps=6, std=1.0):
    torch.manual_seed(0)
    q = (torch.randn(B, nh, L, hd, device="cuda") * std * (hd ** -0.5 * ops.Q_PRESCALE if pre else 1.0)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = (torch.randn(B, nh, Tp, hd, device="cuda") * std).bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    for kk, vv in knobs.items(): ops.set_tuning(kk, vv)
    outs = []
    for r in range(reps):
        out = torch.full((B, L, nh * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, causal, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
        torch.cuda.synchronize(); outs.append(out.clone())
    nd = sum(int((o.view(torch.int16) != outs[0].view(torch.int16)).sum()) for o in outs[1:])
    return nd
V = {"dma": dict(attn_pp=0, attn_il=0), "pp": dict(attn_pp=1, attn_il=0), "il8": dict(attn_pp=1, attn_il=1, attn_il_waves=8), "il4": dict(attn_pp=1, attn_il=1, attn_il_waves=4)}
for (B, L, nh, hd, c, pre, std) in ((1, 2531, 32, 96, True, True, 1.0), (1, 2531, 32, 96, True, True, 4.0), (17, 577, 16, 64, False, False, 1.0), (1, 2531, 32, 96, True, False, 4.0), (1, 8192, 8, 96, True, True, 1.0)):
    for name, kn in V.items():
        if name.startswith("il") and not pre: continue
        print(f"B={B} L={L} hd={hd} causal={c} prescaled={pre} std={std} {name}: differing output words over 5 repeats: {run(B, L, nh, hd, c, pre, kn, std=std)}", flush=True)
