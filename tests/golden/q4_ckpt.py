"""Rebuild, from the seeded synthetic weights, the 4-bit checkpoint that the REFERENCE's `_quantize` wrote when
`gen_golden_refmodel.py` ran (phi_3_vision_mlx.py:291-305) -- and prove it is the same file: `ref_model_tiny.json` holds the name,
dtype, shape and sha256 of every tensor of the reference-written `quantized_model.safetensors`.  Used by tests/test_refmodel.py
(CPU: loader + oracle) and tests/test_model_gpu.py (HIP path) so that neither needs /root/reference."""
import hashlib
import json
import os

import torch


def build(path, case_meta, head_seed, spread=4.0):
    """Write `path`/config.json + quantized_model.safetensors; returns (cfg dict, {name: tensor}) after checking every tensor
    against the reference-written file's record."""
    from safetensors.torch import save_file
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.weights import mlx_quantize, peaked_lm_head, synth_weights
    d = tiny_config_dict(vision=True)
    w = synth_weights(make_config(d), seed=0, std_scale=4.0)
    w["lm_head.weight"] = peaked_lm_head(w["lm_head.weight"], spread, int(head_seed))
    rec = case_meta["tensors"]
    quantised = {k[:-len(".scales")] for k in rec if k.endswith(".scales")}
    out = {}
    for k, v in w.items():
        base = k[:-len(".weight")] if k.endswith(".weight") else None
        if base in quantised:
            out[k], out[base + ".scales"], out[base + ".biases"] = mlx_quantize(v)
        elif "patch_embedding.weight" in k:
            out[k] = v.permute(0, 2, 3, 1).contiguous()               # the reference's `_get_wt` transposes, `sanitized` keeps it so
        else:
            out[k] = v
    assert set(out) == set(rec), (sorted(set(out) ^ set(rec))[:6])
    for k, t in out.items():
        dt, shape, sha = rec[k]
        assert str(t.dtype).replace("torch.", "") == dt and list(t.shape) == shape, (k, t.dtype, t.shape, dt, shape)
        assert hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest() == sha, f"{k}: bytes differ from the reference-written file"
    os.makedirs(path, exist_ok=True)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(path, case_meta["file"]))
    cfg = dict(d, **case_meta["config_flags"])
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    return cfg, out
