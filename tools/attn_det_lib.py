"""Ten launches of the 128-query attention kernel at the ViT's shape with the library given as argv[1] (P3V_LIB): number of
output words that differ from the first launch, per launch.  Used to bisect the non-deterministic plain-Q softmax (round 3)."""
import sys, os, ctypes, torch
sys.path.insert(0, "/root/repo")
os.environ["P3V_LIB"] = sys.argv[1]
from phi_3_vision_mlx_amd import ops
B, L, nh, hd = 17, 577, 16, 64
torch.manual_seed(0)
q = torch.randn(B, nh, L, hd, device="cuda").bfloat16()
Tp = 640
k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16(); v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
ops.set_tuning("attn_pp", 0); ops.set_tuning("attn_il", 0)
outs = []
for r in range(8):
    out = torch.full((B, L, nh * hd), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, False, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=False)
    torch.cuda.synchronize(); outs.append(out.clone())
print(sys.argv[1], "differing words:", [int((o.view(torch.int16) != outs[0].view(torch.int16)).sum()) for o in outs[1:]])
