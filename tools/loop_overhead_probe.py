"""Where does the reference-defined decode rate lose 1-3 % against the bare graph replays?  Host-side stamps around _generate's loop
(api.greedy_loop) on a 2531-token text prompt: python tools/loop_overhead_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import api
from phi_3_vision_mlx_amd.api import load_synthetic

model, processor = load_synthetic(blind_model=True, seed=0, device="cuda:0")
K = 128
ids = torch.randint(3, 32000, (1, 2531), dtype=torch.int64, generator=torch.Generator().manual_seed(0))
logits, cache = model(input_ids=ids, max_tokens=3 * K + 40)
token = model.ops.argmax(logits[:, -1].contiguous())[:, None] if hasattr(model, "ops") else api.model_ops.argmax(logits[:, -1].contiguous())[:, None]
for _ in range(8):
    _, token = model.greedy_step(token, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    _, token = model.greedy_step(token, cache)
torch.cuda.synchronize()
dev = time.perf_counter() - t0
print(f"bare replays: {dev / K * 1e3:.4f} ms/step")
for rep in range(2):
    streamer = api.Streamer(processor, False, True)
    stopper = api.TokenStopper(processor, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    streamer(api._rows(token))
    t1 = time.perf_counter()
    token = api.greedy_loop(model, token, cache, K, streamer, stopper)
    t2 = time.perf_counter()
    _, n = streamer.end()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"loop rep {rep}: first-token read {1e3 * (t1 - t0):.3f} ms | greedy_loop {1e3 * (t2 - t1):.3f} ms = {1e3 * (t2 - t1) / K:.4f} ms/step | "
          f"Streamer.end {1e3 * (t3 - t2):.3f} ms | sync {1e3 * (t4 - t3):.3f} ms | whole span {1e3 * (t4 - t0) / K:.4f} ms/step")
