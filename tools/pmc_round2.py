"""Minimal eager launches for PMC collection, round 2: which = gemv | fp8gemm | attn (no graphs, no weight generation)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
which = sys.argv[1]
if which == "fp8gemm":
    for M, N, K, epi in ((4096, 4096, 4096, ops.EPI_NONE), (2531, 9216, 3072, ops.EPI_NONE), (2531, 8192, 3072, ops.EPI_SILU_MUL)):
        rows = 2 * N if epi == ops.EPI_SILU_MUL else N
        a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device="cuda")
        w8 = [torch.randint(0, 120, (rows, K), dtype=torch.uint8, device="cuda") for _ in range(3)]
        sa, sw = torch.ones(M, device="cuda"), torch.ones(rows, device="cuda")
        for i in range(8):
            ops.gemm_fp8(a8, sa, w8[i % 3], sw, epi)
elif which == "attn":
    B, nh, hd, T, past = 1, 32, 96, 2688, 2540
    qkv = torch.randn((1, 3 * nh * hd), device="cuda").bfloat16()
    kc = [torch.randn((B, nh, T, hd), device="cuda").bfloat16() for _ in range(6)]          # 6 x 33 MB: rotate past the caches
    vc = [torch.randn((B, nh, hd, T), device="cuda").bfloat16() for _ in range(6)]
    cos = torch.ones((B, 1, hd // 2), device="cuda")
    sin = torch.zeros_like(cos)
    out = torch.empty((B, 1, nh * hd), dtype=torch.bfloat16, device="cuda")
    n_split = T // 128
    ws = ops.attention_ws(B, 1, nh, hd, n_split, "cuda")
    for i in range(18):
        ops.attention_decode(qkv, cos, sin, 1, kc[i % 6], vc[i % 6], out, B, 1, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, merge_in_launch=True)
torch.cuda.synchronize()
print("done")
