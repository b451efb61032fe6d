import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tests/golden", "/root/repo/oracle"]
import numpy as np, torch
from test_model_gpu import GOLDEN, HEAVY, head_row_norms, _from_bits
from golden_inputs import vqa_request
from phi_3_vision_mlx_amd.api import load_synthetic
for tiny, tag in ((True, "c2h"), (True, "c5wh"), (True, "c5h"), (False, "c2h"), (False, "c5wh"), (False, "c5h")):
    g = np.load(f"{GOLDEN}/{'tiny_' if tiny else ''}{tag}_oracle.npz")
    model, proc = load_synthetic(tiny=tiny, seed=0, device="cuda:0", std_scale=4.0 if tiny else 1.0, outliers=True,
                                 lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]), **HEAVY[tag])
    inp = vqa_request(proc.img_processor, 0)
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    norms = head_row_norms(model)
    logits, cache = model(**inp, max_tokens=4)
    errs = []
    n_steps = g["tokens"].shape[1]
    for step in range(n_steps):
        ref = _from_bits(g["logits_bf16"][:, step])
        got = logits[:, -1].float().cpu().reshape(ref.shape)
        z = ((got - ref).abs() / norms).max() / (ref / norms).abs().max()
        errs.append(round(100 * z.item(), 2))
        if step + 1 < n_steps:
            logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    print(("tiny " if tiny else "full ") + tag, "z-space logit error per step (%):", errs, "tol", float(g["rel_tol"][0]), flush=True)
    del model, cache
    torch.cuda.empty_cache()
