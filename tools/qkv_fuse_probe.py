"""qkv projection + split / RoPE / KV append: the fused launch (p3v_gemm_qkv) against the two calls, decoder and CLIP shapes, alternated."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit
BF16 = torch.bfloat16
for name, B, L, Lp, nh, hd, K, rot, bias in (("decoder 2531", 1, 2531, 2531, 32, 96, 3072, True, False), ("clip 17 x 584", 17, 584, 584, 16, 64, 1024, False, True)):
    M, N = B * Lp, 3 * nh * hd
    a = torch.randn(M, K, device="cuda").to(BF16)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF16) for _ in range(3)]
    bq = torch.randn(N, device="cuda").to(BF16) if bias else None
    Tp = (Lp + 200 + 127) // 128 * 128
    cos = torch.rand((B, Lp + 200, hd // 2), device="cuda") if rot else None
    sin = torch.rand((B, Lp + 200, hd // 2), device="cuda") if rot else None
    q = torch.empty((B, nh, Lp, hd), dtype=BF16, device="cuda")
    k, v = torch.empty((B, nh, Tp, hd), dtype=BF16, device="cuda"), torch.empty((B, nh, hd, Tp), dtype=BF16, device="cuda")
    qkv = torch.empty((M, N), dtype=BF16, device="cuda")

    def two(i):
        ops.gemm(a, Ws[i], ops.EPI_BIAS if bias else ops.EPI_NONE, bias=bq, out=qkv)
        ops.rope_kv_append(qkv, cos, sin, q, k, v, B, Lp, nh, nh, hd, 0, Tp, True, Lp + 200, 1, q_scale=1.3)

    def one(i):
        assert ops.gemm_qkv(a, Ws[i], cos, sin, q, k, v, B, Lp, nh, nh, hd, 0, Tp, True, Lp + 200, 1, q_scale=1.3, bias=bq)

    def gemm_only(i):
        ops.gemm(a, Ws[i], ops.EPI_BIAS if bias else ops.EPI_NONE, bias=bq, out=qkv)
    t = {"gemm alone": [], "gemm + rope_kv_append": [], "gemm_qkv": []}
    for rep in range(3):
        t["gemm alone"].append(timeit(gemm_only, 3, iters=10))
        t["gemm + rope_kv_append"].append(timeit(two, 3, iters=10))
        t["gemm_qkv"].append(timeit(one, 3, iters=10))
    print(f"{name}: " + "  ".join(f"{n}: {sorted(x)[1] * 1e3:6.1f} us" for n, x in t.items()), flush=True)
