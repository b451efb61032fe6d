import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
x = torch.zeros(64, dtype=torch.int32, device="cuda")
y = torch.zeros(64, dtype=torch.int32, device="cuda")
xb = torch.randn(1, 3072, device="cuda").bfloat16(); w = torch.ones(3072, device="cuda").bfloat16(); yb = torch.empty_like(xb)
def bench(name, fn, n=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = ops.Graph(); g.begin()
        for i in range(n): fn()
        g.end(); g.launch(); s.synchronize()
        a, b = ops.Event(), ops.Event(); a.record(); g.launch(); b.record(); s.synchronize()
        print(f"{name:40s}: {a.elapsed_ms(b)*1e3/n:.2f} us per kernel")
bench("add_i32 same buffer", lambda: ops.add_i32(x, 1))
i = [0]
def alt():
    i[0] ^= 1
    ops.add_i32(x if i[0] else y, 1)
bench("add_i32 alternating buffers", alt)
bench("rmsnorm 1 row", lambda: ops.rmsnorm(xb, w, 1e-5, out=yb))
def mix():
    ops.add_i32(x, 1); ops.rmsnorm(xb, w, 1e-5, out=yb)
bench("add_i32 + rmsnorm (per pair /2)", mix, 100)
