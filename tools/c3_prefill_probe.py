"""config 3 (32768-token text prompt) prefill, fused qkv epilogue on / off, alternated: python tools/c3_prefill_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd.api import load_synthetic

model, _ = load_synthetic(blind_model=True, seed=0, device="cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ids = torch.randint(3, 32000, (1, S), dtype=torch.int64, generator=torch.Generator().manual_seed(0))
for rep in range(4):
    for flag in ("1", "0"):
        os.environ["P3V_QKV_FUSE"] = flag
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits, cache = model(input_ids=ids, max_tokens=136)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        del cache, logits
        print(f"rep {rep} P3V_QKV_FUSE={flag}: prefill {dt:.1f} ms", flush=True)
