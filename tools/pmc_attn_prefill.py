"""A few eager launches of the prompt-sized attention for PMC collection (tools/pmc_attn_prefill.sh): argv = L pp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
L, pp, nh, hd = int(sys.argv[1]), int(sys.argv[2]), 32, 96
q = (torch.randn(1, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
k = torch.randn(1, nh, L, hd, device="cuda").bfloat16()
v = torch.randn(1, nh, hd, L, device="cuda").bfloat16()
out = torch.empty(1, L, nh * hd, device="cuda", dtype=torch.bfloat16)
ops.set_tuning("attn_pp", min(pp, 1))
ops.set_tuning("attn_il", int(pp == 2))                          # pp = 2: k_attn_prefill_il
for _ in range(4):
    ops.attention(q, out, 1, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=L, new_is_cache=True, q_prescaled=True)
torch.cuda.synchronize()
print("done")
