"""Model configuration for the Phi-3-Vision / Phi-3-mini-128K hot path.

Mirrors what the reference reads from ``config.json`` into a SimpleNamespace
(reference phi_3_vision_mlx.py:359-363, keys listed in SURVEY.md App. B).
The rope factor arrays are the Phi-3-vision-128k-instruct values printed in the
reference's own notebook (assets/su_rope_explained.ipynb, cell 9); a real
``config.json`` always overrides them.
"""
import json
import math
from types import SimpleNamespace

SHORT_FACTOR = [
    1.05, 1.05, 1.05, 1.1, 1.1, 1.1, 1.2500000000000002, 1.2500000000000002,
    1.4000000000000004, 1.4500000000000004, 1.5500000000000005, 1.8500000000000008,
    1.9000000000000008, 2.000000000000001, 2.000000000000001, 2.000000000000001,
    2.000000000000001, 2.000000000000001, 2.000000000000001, 2.000000000000001,
    2.000000000000001, 2.000000000000001, 2.000000000000001, 2.000000000000001,
    2.000000000000001, 2.000000000000001, 2.000000000000001, 2.000000000000001,
    2.000000000000001, 2.000000000000001, 2.000000000000001, 2.000000000000001,
    2.1000000000000005, 2.1000000000000005, 2.2, 2.3499999999999996,
    2.3499999999999996, 2.3499999999999996, 2.3499999999999996, 2.3999999999999995,
    2.3999999999999995, 2.6499999999999986, 2.6999999999999984, 2.8999999999999977,
    2.9499999999999975, 3.049999999999997, 3.049999999999997, 3.049999999999997,
]
LONG_FACTOR = [
    1.0299999713897705, 1.0499999523162842, 1.0499999523162842, 1.0799999237060547,
    1.2299998998641968, 1.2299998998641968, 1.2999999523162842, 1.4499999284744263,
    1.5999999046325684, 1.6499998569488525, 1.8999998569488525, 2.859999895095825,
    3.68999981880188, 5.419999599456787, 5.489999771118164, 5.489999771118164,
    9.09000015258789, 11.579999923706055, 15.65999984741211, 15.769999504089355,
    15.789999961853027, 18.360000610351562, 21.989999771118164, 23.079999923706055,
    30.009998321533203, 32.35000228881836, 32.590003967285156, 35.56000518798828,
    39.95000457763672, 53.840003967285156, 56.20000457763672, 57.95000457763672,
    59.29000473022461, 59.77000427246094, 59.920005798339844, 61.190006256103516,
    61.96000671386719, 62.50000762939453, 63.3700065612793, 63.48000717163086,
    63.48000717163086, 63.66000747680664, 63.850006103515625, 64.08000946044922,
    64.760009765625, 64.80001068115234, 64.81001281738281, 64.81001281738281,
]

# CLIP ViT-L/14-336 is hard-coded in the reference (phi.py:375-384).
CLIP_L_336 = dict(hidden_size=1024, image_size=336, intermediate_size=4096,
                  layer_norm_eps=1e-05, num_attention_heads=16, num_channels=3,
                  num_hidden_layers=24, patch_size=14)


def phi3v_config_dict(vision=True):
    """Full-size Phi-3-Vision (or blind Phi-3-mini-128K) config as a dict."""
    d = dict(
        architectures=["Phi3VForCausalLM" if vision else "Phi3ForCausalLM"],
        hidden_size=3072, num_attention_heads=32, num_key_value_heads=32,
        num_hidden_layers=32, intermediate_size=8192, vocab_size=32064,
        rms_norm_eps=1e-05, rope_theta=10000.0, max_position_embeddings=131072,
        original_max_position_embeddings=4096,
        rope_scaling=dict(type="su", short_factor=list(SHORT_FACTOR), long_factor=list(LONG_FACTOR)),
    )
    if vision:
        d["img_processor"] = dict(image_dim_out=1024, name="clip_vision_model", num_img_tokens=144)
        d["clip"] = dict(CLIP_L_336)
    return d


def tiny_config_dict(vision=True):
    """A tiny config with the same structure (head_dim stays 96 so the same
    kernels are exercised; CLIP head_dim stays 64; image stays 336/14 so the
    reference's HD transform and token-count formula apply unchanged)."""
    n = 48
    d = dict(
        architectures=["Phi3VForCausalLM" if vision else "Phi3ForCausalLM"],
        hidden_size=192, num_attention_heads=2, num_key_value_heads=2,
        num_hidden_layers=2, intermediate_size=256, vocab_size=32064,
        rms_norm_eps=1e-05, rope_theta=10000.0, max_position_embeddings=131072,
        original_max_position_embeddings=4096,
        rope_scaling=dict(type="su", short_factor=list(SHORT_FACTOR[:n]), long_factor=list(LONG_FACTOR[:n])),
    )
    if vision:
        d["img_processor"] = dict(image_dim_out=128, name="clip_vision_model", num_img_tokens=144)
        d["clip"] = dict(hidden_size=128, image_size=336, intermediate_size=256,
                         layer_norm_eps=1e-05, num_attention_heads=2, num_channels=3,
                         num_hidden_layers=3, patch_size=14)
    return d


def make_config(d=None, **kwargs):
    """dict (+ kwargs overlay, as reference `_get_cfg`) -> SimpleNamespace."""
    d = dict(d if d is not None else phi3v_config_dict())
    d.update(kwargs)
    cfg = SimpleNamespace(**d)
    if getattr(cfg, "img_processor", None) is not None and not hasattr(cfg, "clip"):
        cfg.clip = dict(CLIP_L_336)
    return cfg


def load_config(json_path, **kwargs):
    """Same error behaviour as reference `_get_cfg` (phi_3_vision_mlx.py:359-369)."""
    try:
        with open(json_path, "r") as f:
            d = json.load(f)
    except FileNotFoundError:
        raise FileNotFoundError(f"Configuration file not found: {json_path}")
    except json.JSONDecodeError:
        raise ValueError(f"Invalid JSON in configuration file: {json_path}")
    return make_config(d, **kwargs)


def is_vision(cfg):
    return cfg.architectures[0].startswith("Phi3V")


def head_dim(cfg):
    return cfg.hidden_size // cfg.num_attention_heads


def rope_scaling_factor(cfg):
    """sqrt(1 + ln(max/orig)/ln(orig)) (reference phi.py:491)."""
    return math.sqrt(1 + math.log(cfg.max_position_embeddings / cfg.original_max_position_embeddings)
                     / math.log(cfg.original_max_position_embeddings))
