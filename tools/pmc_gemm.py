"""Minimal eager launches of the prefill GEMM shapes for PMC collection (MFMA busy, LDS bank conflicts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
M, H, I = 2531, 3072, 8192
x = torch.randn((M, H), device="cuda").to(torch.bfloat16)
w_gu = torch.randn((2 * I, H), device="cuda").to(torch.bfloat16)
w_qkv = torch.randn((3 * H, H), device="cuda").to(torch.bfloat16)
sq = torch.randn((4096, 4096), device="cuda").to(torch.bfloat16)
bias = torch.zeros(4096, device="cuda").to(torch.bfloat16)
for i in range(6):
    ops.gemm(x, w_gu, ops.EPI_SILU_MUL)     # -> k_gemm256<6,..> on rows [0,2048) + k_gemm<6,..> on the rest (round packing)
    ops.gemm(x, w_qkv, ops.EPI_NONE)         # -> k_gemm256<0,..> on rows [0,1792) + k_gemm<0,..>
    ops.gemm(sq, sq, ops.EPI_BIAS, bias=bias)  # 4096^3, exactly one round of 256 big tiles: k_gemm256<1,..>
torch.cuda.synchronize()
print("done")
