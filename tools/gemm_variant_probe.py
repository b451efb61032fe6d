import os, sys
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{sys.argv[1]}.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit
ops.set_tuning("gemm_big_rows", 1 << 20)
for M, N, K in ((1792, 9216, 3072), (1792, 4096, 1024), (4096, 4096, 4096)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(4)]
    ms = timeit(lambda i: ops.gemm(A, Ws[i], ops.EPI_NONE), 4)
    print(f"{sys.argv[1]:12s} M={M:5d} N={N:5d} K={K:5d}: {ms*1e3:7.1f} us ({2*M*N*K/ms/1e9:6.0f} TF/s)", flush=True)
