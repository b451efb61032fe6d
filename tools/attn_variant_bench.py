"""Kernel time of k_attn_prefill_pp builds (tools/build_variant.sh <name> p3v_attention.hip -D...) at one shape; timing
experiments only (several variants compute garbage)."""
import os, sys
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{sys.argv[1]}.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
L, nh, hd = int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 32, 96
pp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
q = (torch.randn(1, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
k = torch.randn(1, nh, L, hd, device="cuda").bfloat16()
v = torch.randn(1, nh, hd, L, device="cuda").bfloat16()
out = torch.empty(1, L, nh * hd, device="cuda", dtype=torch.bfloat16)
ops.set_tuning("attn_pp", pp)
f = lambda: ops.attention(q, out, 1, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=L, new_is_cache=True, q_prescaled=True)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = ops.Event(), ops.Event()
a.record()
for _ in range(10): f()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_ms(b) / 10
print(f"{sys.argv[1]:20s} L={L} pp={pp}: {ms * 1e3:8.1f} us  {2 * nh * L * L * hd / ms / 1e9:7.1f} TF/s", flush=True)
