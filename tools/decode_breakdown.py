"""Per-op time of ONE eager decode step at the bench context (HIP events around every op launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model, _ = load_synthetic(blind_model=True, device="cuda:0")
ids = np.random.default_rng(0).integers(3, 32000, (B, ctx))
lg, cache = model(input_ids=ids, max_tokens=16)
tok = ops.argmax(lg[:, -1].contiguous())[:, None]
for _ in range(2): lg, cache = model(input_ids=tok, cache=cache)
recs = []
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        e0, e1 = ops.Event(), ops.Event()
        e0.record(); r = orig(*a, **k); e1.record()
        tag = name
        if name == "gemv":
            w = a[1]; tag = f"gemv N={w.shape[0]} K={w.shape[1]}"
        recs.append((tag, e0, e1)); return r
    setattr(ops, name, f); return orig
names = ["gemv", "gemm", "rope_kv_append", "attention", "attention_decode", "embed_gather", "rmsnorm", "argmax"]
origs = {n: wrap(n) for n in names}
for rep in range(3):
    recs.clear()
    lg, cache = model(input_ids=tok, cache=cache)
    torch.cuda.synchronize()
for n, o in origs.items(): setattr(ops, n, o)
agg = collections.OrderedDict()
for tag, a, b in recs:
    agg.setdefault(tag, []).append(a.elapsed_ms(b) * 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"context {ctx} B={B}: sum of op times {tot/1e3:.3f} ms")
for tag, v in agg.items():
    print(f"  {tag:28s} n={len(v):3d} avg {np.mean(v):8.2f} us  total {sum(v)/1e3:7.3f} ms")
# graph replay time for reference
import time
t = tok
for _ in range(3): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); print(f"graph replay: {(time.perf_counter()-t0)/8*1e3:.3f} ms/step")
