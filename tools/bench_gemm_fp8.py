"""bf16 vs W8A8 (fp8 MFMA) prefill projections at the decoder's shapes, M = 2531 (run on the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
for name, N, K, epi in (("qkv", 9216, 3072, ops.EPI_NONE), ("o_proj", 3072, 3072, ops.EPI_RESID_BF16),
                        ("gate_up", 8192, 3072, ops.EPI_SILU_MUL), ("down", 3072, 8192, ops.EPI_RESID_BF16),
                        ("square 4096", 4096, 4096, ops.EPI_NONE), ("square 8192", 8192, 8192, ops.EPI_NONE)):
    m = M if "square" not in name else N
    rows = 2 * N if epi == ops.EPI_SILU_MUL else N
    nrot = 4
    A = torch.randn(m, K, device="cuda").bfloat16()
    Ws = [torch.randn(rows, K, device="cuda").bfloat16() * 0.02 for _ in range(nrot)]
    W8 = [ops.quantize_fp8_rows(w) for w in Ws]
    res = torch.zeros(m, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=res, out=res) if epi == ops.EPI_RESID_BF16 else {}
    nw = torch.ones(K, device="cuda").bfloat16()
    t_bf = timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), nrot)
    a8, sa = ops.quant_fp8_rows(A, nw, 1e-5)
    t_q = timeit(lambda i: ops.quant_fp8_rows(A, nw, 1e-5, out=a8, scale=sa), nrot)
    t_f8 = timeit(lambda i: ops.gemm_fp8(a8, sa, W8[i][0], W8[i][1], epi, **kw), nrot)
    fl = 2.0 * m * rows * K
    print(f"{name:12s} M={m:5d} N={rows:6d} K={K:5d}: bf16 {t_bf*1e3:7.1f} us ({fl/t_bf/1e9:7.1f} TF/s)   fp8 {t_f8*1e3:7.1f} us ({fl/t_f8/1e9:7.1f} TF/s)"
          f"   quantise {t_q*1e3:5.1f} us   speed-up incl. quantiser {t_bf/(t_f8+t_q):.2f}x")
