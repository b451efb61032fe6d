"""Prefill wall time of a text prompt (median of 5) for the weight formats: python tools/prefill_time.py [ctx] (P3V_INT4=1 / P3V_FP8=1 P3V_QCACHE=1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
model, _ = load_synthetic(blind_model=True, device="cuda:0", use_quantized_cache=bool(os.environ.get("P3V_QCACHE")), quantized_fp8=bool(os.environ.get("P3V_FP8")), quantized_int4=bool(os.environ.get("P3V_INT4")))
ids = np.random.default_rng(0).integers(3, 32000, (1, ctx))
res = []
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lg, cache = model(input_ids=ids, max_tokens=16)
    t = ops.argmax(lg[:, -1].contiguous())
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) * 1e3)
    del cache
print(f"ctx {ctx} env {dict((k, v) for k, v in os.environ.items() if k.startswith('P3V_'))}: prefill median {sorted(res[1:])[2]:.2f} ms  (reps {[round(r, 2) for r in res]})")
