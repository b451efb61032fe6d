"""prefill + N graph-replayed greedy steps (for rocprofv3 tracing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
model, _ = load_synthetic(blind_model=True, device="cuda:0", use_quantized_cache=bool(os.environ.get("P3V_QCACHE")),
                          quantized_fp8=bool(os.environ.get("P3V_FP8")),   # config 5: P3V_FP8=1 P3V_QCACHE=1
                          quantized_int4=bool(os.environ.get("P3V_INT4")))  # MLX 4-bit weights: P3V_INT4=1
ids = np.random.default_rng(0).integers(3, 32000, (B, ctx))
lg, cache = model(input_ids=ids, max_tokens=max(steps + 4, 144))   # >= bench.py's cache capacity (8 + 128 + 8): same tile count
t = ops.argmax(lg[:, -1].contiguous())[:, None]
for _ in range(steps): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize()
