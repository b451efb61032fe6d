"""Forty launches of the prompt attention with the launcher's own kernel choice at seven shapes (the ViT, config 2, batches,
short, long): output words that differ from the first launch (0 everywhere on the final tree)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from phi_3_vision_mlx_amd import ops
def run(B, L, nh, hd, causal, pre, reps=40):
    torch.manual_seed(1)
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE if pre else 1.0)).bfloat16()
    Tp = (L + 63) // 64 * 64
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    first, nd = None, 0
    for r in range(reps):
        out = torch.full((B, L, nh * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, causal, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
        torch.cuda.synchronize()
        if first is None: first = out.clone()
        else: nd += int((out.view(torch.int16) != first.view(torch.int16)).sum())
    return nd
for c in ((17, 577, 16, 64, False, False), (1, 2531, 32, 96, True, True), (4, 2531, 32, 96, True, True), (8, 512, 32, 96, True, True), (1, 128, 32, 96, True, True), (1, 8192, 32, 96, True, True), (2, 5000, 32, 96, True, True)):
    print(c, "launcher's choice, 40 launches, differing words:", run(*c), flush=True)
