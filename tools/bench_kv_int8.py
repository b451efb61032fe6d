"""Decode speed, bf16 KV vs int8 KV (quantize_cache=True), full-size text model (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd.api import load_synthetic

cases = [(1, 2531), (1, 8192), (1, 32768), (8, 2531), (8, 8192)]
res = {}
for q in (False, True):
    model, _ = load_synthetic(blind_model=True, seed=0, device="cuda:0", use_quantized_cache=q)
    for B, S in cases:
        ids = np.random.default_rng(0).integers(3, 32000, (B, S)).astype(np.int64)
        tok, cache = model.greedy_prefill(80, input_ids=ids)
        for _ in range(8):
            _, tok = model.greedy_step(tok, cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(48):
            _, tok = model.greedy_step(tok, cache)
        torch.cuda.synchronize()
        res[(q, B, S)] = (time.perf_counter() - t0) / 48 * 1e3
        del cache
        torch.cuda.empty_cache()
    del model
    torch.cuda.empty_cache()
for B, S in cases:
    a, b = res[(False, B, S)], res[(True, B, S)]
    print(f"B={B} ctx={S:6d}: bf16 KV {a:7.3f} ms/step ({B/a*1e3:7.1f} tok/s)   int8 KV {b:7.3f} ms/step ({B/b*1e3:7.1f} tok/s)   int8/bf16 time {b/a:.3f}")
