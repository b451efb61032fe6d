"""THIRD-PARTY cross-check of the oracle's model math: Hugging Face `transformers` (5.15 in this image) Phi3ForCausalLM
(longrope) and CLIPVisionModel, fp32, on the same seeded synthetic weights.

This is NOT the MLX reference (which cannot run here) -- it corroborates the parts of the restatement that
the two model families share: fused qkv / gate_up split order, half-split RoPE with the Su/LongRoPE factors and the
sqrt(1 + ln(s)/ln(L)) magnitude, pre-norm residual layout, causal + padding mask, KV-cache decode; CLIP embeddings,
pre-LN encoder blocks with quick-GELU, hidden_states[-2] == "all layers but the last, no post-LN" (phi.py:219).
Reference quirks that HF does NOT share are outside this check and stay covered by the audited restatement only:
Q1 (fp32 attention by dtype promotion under bf16 weights -- both sides run pure fp32 here), Q2 (the short/long choice
is made once from S + max_tokens in the reference, per forward from max(position)+1 in HF: the cases below keep both on
the same side), Q4 (HD merge order; Phi-3-V's image embedding is not in `transformers`), Q6 (second BOS), Q7 (pad rows).

    python tests/golden/gen_golden_hf.py      -> tests/golden/hf_crosscheck.npz (HF outputs; the test compares the oracle
                                                 with them, and with a live HF run when `transformers` is importable)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)
from phi_3_vision_mlx_amd.config import make_config, phi3v_config_dict, tiny_config_dict  # noqa: E402
from phi_3_vision_mlx_amd.weights import synth_weights  # noqa: E402

V_PREFIX = "model.vision_embed_tokens.img_processor.vision_model."


def text_config(kind):
    """tiny | tiny_long (original window 32 -> long factors at S = 50) | wide2 (2 layers at the full 3072 width)."""
    if kind == "wide2":
        d = phi3v_config_dict(vision=False)
        d["num_hidden_layers"] = 2
    else:
        d = tiny_config_dict(vision=False)
        if kind == "tiny_long":
            d["original_max_position_embeddings"] = 32
    return make_config(d)


def hf_phi3(cfg, w):
    from transformers import Phi3Config, Phi3ForCausalLM
    hc = Phi3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                    num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                    num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                    max_position_embeddings=cfg.max_position_embeddings,
                    original_max_position_embeddings=cfg.original_max_position_embeddings,
                    rope_scaling={"type": "longrope", "short_factor": cfg.rope_scaling["short_factor"],
                                  "long_factor": cfg.rope_scaling["long_factor"]},
                    attention_dropout=0.0, tie_word_embeddings=False, pad_token_id=0, attn_implementation="eager")
    m = Phi3ForCausalLM(hc).float().eval()
    m.load_state_dict({k: v.float() for k, v in w.items()}, strict=True)
    return m


def hf_clip(cfg, w):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = cfg.clip
    hc = CLIPVisionConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                          num_attention_heads=c["num_attention_heads"], image_size=c["image_size"], patch_size=c["patch_size"],
                          hidden_act="quick_gelu", layer_norm_eps=c["layer_norm_eps"], attn_implementation="eager")
    m = CLIPVisionModel(hc).float().eval()
    pre = "vision_model." if any(k.startswith("vision_model.") for k in m.state_dict()) else ""      # differs across versions
    sd = {pre + k[len(V_PREFIX):]: v.float() for k, v in w.items() if k.startswith(V_PREFIX)}
    m.load_state_dict(sd, strict=True)
    return m


def text_cases(kind):
    rng = np.random.default_rng({"tiny": 1, "tiny_long": 2, "wide2": 3}[kind])
    S = 50 if kind != "wide2" else 40
    ids = rng.integers(3, 32000, (1, S)).astype(np.int64)
    return ids


def clip_pixels(n, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal((n, 3, 336, 336)).astype(np.float32))


def main():
    out = {}
    with torch.no_grad():
        for kind in ("tiny", "tiny_long", "wide2"):
            cfg = text_config(kind)
            w = synth_weights(cfg, seed=0, std_scale=4.0 if kind != "wide2" else 1.0)
            m = hf_phi3(cfg, w)
            ids = text_cases(kind)
            lg = m(input_ids=torch.as_tensor(ids)).logits
            out[f"{kind}_last_logits"] = lg[0, -1].numpy()
            out[f"{kind}_mid_logits_sub"] = lg[0, ::7, ::97].numpy()              # a grid over positions x vocabulary
            print(kind, "HF |logit|max", lg.abs().max().item())
        cfg = make_config(tiny_config_dict(vision=True))
        w = synth_weights(cfg, seed=0, std_scale=4.0)
        hs = hf_clip(cfg, w)(pixel_values=clip_pixels(2), output_hidden_states=True).hidden_states[-2]
        out["clip_tiny_feats_sub"] = hs[:, 1:, :][:, ::5].numpy()
        print("clip tiny |feat|max", hs.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "hf_crosscheck.npz"), **out)
    print("wrote hf_crosscheck.npz")


if __name__ == "__main__":
    main()
