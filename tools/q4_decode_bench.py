"""B = 1 decode on MLX 4-bit group-64 weights (the reference's quantize_model format) at BASELINE config 2's context:
ms per step as bare graph replays.  python tools/q4_decode_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd.api import load_synthetic
from phi_3_vision_mlx_amd import ops
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model, processor = load_synthetic(blind_model=True, seed=0, device="cuda:0", quantized_int4=True)
ids = np.random.default_rng(0).integers(1000, 30000, (1, 2531))
logits, cache = model(input_ids=ids, max_tokens=2 * steps + 24)
token = ops.argmax(logits[:, -1, :].contiguous())[:, None]
for _ in range(8):
    _, token = model.greedy_step(token, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    _, token = model.greedy_step(token, cache)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("4-bit weights, 2531-token context, fuse=%s: %.4f ms per step = %.1f tok/s" % (os.environ.get("P3V_ATTN_FUSE_OPROJ", "1"), dt * 1e3, 1 / dt))
