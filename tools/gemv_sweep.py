import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit
K = 3072
x = torch.randn(1, K, device="cuda").bfloat16()
nw = torch.ones(K, device="cuda").bfloat16()
for N in (64, 512, 3072, 9216, 18432, 32064, 65536, 131072):
    nrot = min(64, max(2, int(700e6 // (N * K * 2)) + 1))
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(nrot)]
    for norm in (False, True):
        kw = dict(norm_w=nw, norm_eps=1e-5) if norm else {}
        ms = timeit(lambda i: ops.gemv(x, Ws[i], ops.EPI_NONE, **kw), nrot, iters=100)
        print(f"N={N:7d} norm={int(norm)} {ms*1e3:8.2f} us  {N*K*2/ms/1e6:8.1f} GB/s")
# empty-kernel floor: tiny rmsnorm
y = torch.empty_like(x)
ms = timeit(lambda i: ops.rmsnorm(x, nw, 1e-5, out=y), 1, iters=200); print(f"rmsnorm 1 row: {ms*1e3:.2f} us (launch floor)")
g = ops.Graph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.begin()
    for i in range(100): ops.rmsnorm(x, nw, 1e-5, out=y)
    g.end()
    g.launch(); s.synchronize()
    a, b = ops.Event(), ops.Event(); a.record(); g.launch(); b.record(); s.synchronize()
    print(f"graph of 100 tiny kernels: {a.elapsed_ms(b)*10:.2f} us per kernel")
    W = torch.randn(3072, K, device="cuda").bfloat16()
    Ws = [torch.randn(3072, K, device="cuda").bfloat16() for _ in range(40)]
    g2 = ops.Graph(); g2.begin()
    for i in range(120): ops.gemv(x, Ws[i % 40], ops.EPI_NONE)
    g2.end(); g2.launch(); s.synchronize()
    a.record(); g2.launch(); b.record(); s.synchronize()
    print(f"graph of 120 o_proj-sized gemv: {a.elapsed_ms(b)*1e3/120:.2f} us per kernel")
