import sys; sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tests/golden", "/root/repo/oracle"]
import numpy as np, torch
from phi_3_vision_mlx_amd.api import load_synthetic
model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0")
S = 32768
ids = np.random.default_rng(4).integers(3, 32000, (1, S + 1)).astype(np.int64)
a, cache = model(input_ids=ids[:, :S], max_tokens=4)
b, _ = model(input_ids=ids[:, S:], cache=cache)
del cache; torch.cuda.empty_cache()
c, _ = model(input_ids=ids, max_tokens=1)
got, ref = b[:, -1].float().cpu(), c[:, -1].float().cpu()
err = (got - ref).abs()
print("32k decode vs prefill: max err", err.max().item(), "|ref|max", ref.abs().max().item(), "rel", (err.max() / ref.abs().max()).item(),
      "frac > 2% of max:", (err > 0.02 * ref.abs().max()).float().mean().item())
