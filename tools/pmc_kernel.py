"""Minimal eager launches of the dominant decode kernel for PMC collection (no graphs, no weight generation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
I, H = 8192, 3072
Ws = [torch.zeros((2 * I, H), dtype=torch.bfloat16, device="cuda") for _ in range(4)]      # 4 x 100 MB > 256 MiB Infinity Cache
x = torch.ones((1, H), dtype=torch.bfloat16, device="cuda")
nw = torch.ones((H,), dtype=torch.bfloat16, device="cuda")
out = torch.empty((1, I), dtype=torch.bfloat16, device="cuda")
for i in range(12):
    ops.gemv(x, Ws[i % 4], ops.EPI_SILU_MUL, norm_w=nw, norm_eps=1e-5, out=out)
torch.cuda.synchronize()
print("done")
