"""Two processes decoding on ONE GPU, neither told about the other: the fused attention + o_proj launch of each can be starved by the
other's kernels; _generate's loop must notice, re-plan and deliver the same tokens as a solo run.
  python tools/cotenant_probe.py solo; (python tools/cotenant_probe.py a & python tools/cotenant_probe.py b; wait)"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import api, ops
from phi_3_vision_mlx_amd.api import load_synthetic

tag = sys.argv[1] if len(sys.argv) > 1 else "solo"
model, processor = load_synthetic(blind_model=True, seed=0, device="cuda:0", lm_head_spread=4.0, lm_head_seed=1)
ids = torch.randint(3, 32000, (1, 2000), dtype=torch.int64, generator=torch.Generator().manual_seed(0))
rows = []
t0 = time.time()
logits, cache = model(input_ids=ids, max_tokens=420)
token = ops.argmax(logits[:, -1].contiguous())[:, None]
rows.append(api._rows(token))
out = api.greedy_loop(model, token, cache, 400, lambda r: rows.append(list(r)), lambda r: False)
torch.cuda.synchronize()
h = hashlib.sha256(str(rows).encode()).hexdigest()[:16]
print(f"{tag}: {len(rows)} tokens, sha {h}, degraded to separate launches: {model.serving}, {time.time() - t0:.1f} s", flush=True)
