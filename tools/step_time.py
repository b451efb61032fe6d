"""Graph-replayed decode step wall time (no profiler) for quick A/B of kernel variants via env vars."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model, _ = load_synthetic(blind_model=True, device="cuda:0", use_quantized_cache=bool(os.environ.get("P3V_QCACHE")), quantized_fp8=bool(os.environ.get("P3V_FP8")), quantized_int4=bool(os.environ.get("P3V_INT4")))
ids = np.random.default_rng(0).integers(3, 32000, (B, ctx))
lg, cache = model(input_ids=ids, max_tokens=int(sys.argv[3]) if len(sys.argv) > 3 else 80)
t = ops.argmax(lg[:, -1].contiguous())[:, None]
for _ in range(8): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): lg, t = model.greedy_step(t, cache)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"ctx {ctx} B {B} env {dict((k, v) for k, v in os.environ.items() if k.startswith('P3V_'))}: {min(res):.3f} ms/step  ({B/min(res)*1e3:.1f} tok/s)")
