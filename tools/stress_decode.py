"""Determinism / race stress of the graph-replayed decode step (in-launch split-KV merge included): two identical runs of
N greedy steps must produce identical tokens and logits, and no NaN may appear (the merge poisons its output on a timeout)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2531
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
model, _ = load_synthetic(blind_model=True, device="cuda:0", use_quantized_cache=bool(os.environ.get("P3V_QCACHE")),
                          quantized_fp8=bool(os.environ.get("P3V_FP8")))
ids = np.random.default_rng(0).integers(3, 32000, (B, ctx))
runs = []
for rep in range(2):
    lg, cache = model(input_ids=ids, max_tokens=steps + 8)
    t = ops.argmax(lg[:, -1].contiguous())[:, None]
    toks, last = [], None
    for i in range(steps):
        lg, t = model.greedy_step(t, cache)
        toks.append(t.clone())
        if i % 100 == 99:
            assert torch.isfinite(lg.float()).all(), f"non-finite logits at step {i}"
    torch.cuda.synchronize()
    runs.append((torch.cat(toks, 1).cpu(), lg.float().cpu().clone()))
    del cache
same_tok = torch.equal(runs[0][0], runs[1][0]); same_lg = torch.equal(runs[0][1], runs[1][1])
print(f"ctx {ctx}, B {B}, {steps} steps x 2 runs: tokens identical {same_tok}, final logits identical {same_lg}, distinct tokens {runs[0][0].unique().numel()}")
assert same_tok and same_lg
