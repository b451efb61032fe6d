"""Minimal eager launches for the round-5 PMC passes: which = attn_o | gemm.
attn_o: the SHIPPED decode attention -- k_attn_decode128_o: split-KV attention + in-launch merge + o_proj + residual (B = L = 1, 2541 keys
        in a 2688-key cache, 32 heads x 96) -- rotating over 6 caches and 6 W_o so that nothing is served from the Infinity Cache.
gemm:   the prefill's dominant GEMM (gate_up, 2531 x 8192 x 3072, SiLU epilogue; big tiles + remainder launch as shipped)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
which = sys.argv[1]
if which == "attn_o":
    B, nh, hd, T, past, H = 1, 32, 96, 2688, 2540, 3072
    qkv = torch.randn((1, 3 * nh * hd), device="cuda").bfloat16()
    kc = [torch.randn((B, nh, T, hd), device="cuda").bfloat16() for _ in range(6)]
    vc = [torch.randn((B, nh, hd, T), device="cuda").bfloat16() for _ in range(6)]
    wo = [(torch.randn((H, nh * hd), device="cuda") * 0.02).bfloat16() for _ in range(6)]
    x = torch.zeros((1, H), device="cuda").bfloat16()
    cos, sin = torch.ones((B, 1, hd // 2), device="cuda"), torch.zeros((B, 1, hd // 2), device="cuda")
    n_split = T // 128
    assert ops.attention_decode_can_fuse_oproj(B, 1, nh, hd, n_split, T, H, True)
    ws = ops.attention_ws(B, 1, nh, hd, n_split, "cuda")
    o = [torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(torch.bfloat16) for _ in range(2)]
    for i in range(18):
        ops.attention_decode(qkv, cos, sin, 1, kc[i % 6], vc[i % 6], o[i & 1], B, 1, nh, nh, hd, hd ** -0.5, past, T, ws, n_split,
                             merge_in_launch=True, o_proj_w=wo[i % 6], o_proj_x=x, o_rearm=o[(i + 1) & 1])
elif which == "gemm":
    M, N, K = 2531, 8192, 3072
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [(torch.randn(2 * N, K, device="cuda") * 0.02).bfloat16() for _ in range(3)]
    for i in range(10):
        ops.gemm(A, Ws[i % 3], ops.EPI_SILU_MUL)
torch.cuda.synchronize()
print("done")
