"""Time to first token of text prompts by length (B = 1, bf16): prefill + argmax + sync, median of 5 after 2 warm-ups (the second sighting
of a length captures its graph where that applies).  python tools/ttft_probe.py [lengths...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic

model, _ = load_synthetic(blind_model=True, seed=0, device="cuda:0")
lengths = [int(a) for a in sys.argv[1:]] or [17, 64, 128, 192, 256, 257, 320, 384, 512, 768, 1024, 1536, 2048]
for S in lengths:
    ids = torch.randint(3, 32000, (1, S), dtype=torch.int64, generator=torch.Generator().manual_seed(S))
    ts = []
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits, cache = model(input_ids=ids, max_tokens=64)
        tok = ops.argmax(logits[:, -1].contiguous()).tolist()
        ts.append((time.perf_counter() - t0) * 1e3)
        del cache, logits
    print(f"S = {S:5d}: {np.median(ts[2:]):7.2f} ms   (reps {[round(t, 2) for t in ts]})", flush=True)
