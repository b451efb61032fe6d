"""The heavy-tailed fixtures' logit error (in units of the fixture's tolerance) under each prompt-attention kernel choice, twice
each: how much of the HIP - oracle difference is which correct kernel happened to run (chaotic amplification)?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")]
import numpy as np, torch
from golden_inputs import vqa_request
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
from test_model_gpu import HEAVY, GOLDEN, head_row_norms, _from_bits
VARIANTS = {"dma": dict(attn_pp=0, attn_il=0), "pp": dict(attn_pp=1, attn_il=0), "il8": dict(attn_pp=-1, attn_il=1, attn_il_waves=8),
            "il4": dict(attn_pp=-1, attn_il=1, attn_il_waves=4), "auto": dict(attn_pp=-1, attn_il=-1, attn_il_waves=-1)}
for tag in sys.argv[1:] or ("c2h", "c5wh", "c5h"):
    g = np.load(f"{GOLDEN}/{tag}_oracle.npz")
    model, proc = load_synthetic(tiny=False, seed=0, device="cuda:0", std_scale=1.0, outliers=True, lm_head_spread=float(g["spread"][0]),
                                 lm_head_seed=int(g["head_seed"][0]), **HEAVY[tag])
    inp = vqa_request(proc.img_processor, 0)
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    n, norms = ref_tok.shape[1], head_row_norms(model)
    rel_tol = torch.as_tensor(g["rel_tol"]).float()
    for name, knobs in VARIANTS.items():
        for rep in range(2):
            for k, v in knobs.items(): ops.set_tuning(k, v)
            logits, cache = model(**inp, max_tokens=n)
            row = []
            for step in range(n):
                ref = _from_bits(g["logits_bf16"][:, step]); got = logits[:, -1].float().cpu().reshape(ref.shape)
                E = (rel_tol * (ref / norms).abs().amax(-1))[:, None]
                row.append((((got - ref).abs() - 2.0 ** -7 * ref.abs()).clamp_min(0) / (E * norms)).max().item())
                if step + 1 < n: logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
            print(f"{tag} (rel_tol {float(rel_tol[0]):.3f}) {name:5s} run {rep}: error / tolerance per step " + " ".join(f"{x:.2f}" for x in row), flush=True)
            del cache
    del model
    torch.cuda.empty_cache()
