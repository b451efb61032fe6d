"""The oracle's model math against a THIRD PARTY: Hugging Face `transformers` Phi3ForCausalLM (longrope) and
CLIPVisionModel, pure fp32 on both sides (the oracle turns into plain fp32 math when it is handed fp32 weights: every
`_linear` then promotes to fp32, as MLX's dtype promotion would).  Not the MLX reference -- see
tests/golden/gen_golden_hf.py for what this does and does not corroborate (reference quirks Q1/Q2/Q4/Q6/Q7 are outside it).
Compared twice: with the committed HF outputs (tests/golden/hf_crosscheck.npz) and with a live HF run."""
import os
import sys

import numpy as np
import pytest
import torch

import phi3v_oracle as orc
from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
from phi_3_vision_mlx_amd.weights import synth_weights

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gen_golden_hf as hf  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_crosscheck.npz"))
TOL = 2e-4          # fp32 on both sides: differences are summation order only (measured: 1e-5 at |logit|max ~ 5)


def f32(w):
    return {k: v.float() for k, v in w.items()}


def close(a, b, tol=TOL):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("kind", ["tiny", "tiny_long", "wide2"])
def test_decoder_matches_hf_phi3(kind):
    """Prefill logits at every position + KV-cache decode steps + a left-padded batch, oracle vs HF Phi3ForCausalLM:
    tiny (short factors), tiny_long (original window 32: LONG factors + the same magnitude factor), wide2 (two layers at
    the full 3072 / 32-head / 8192 width)."""
    cfg = hf.text_config(kind)
    w = synth_weights(cfg, seed=0, std_scale=4.0 if kind != "wide2" else 1.0)
    o = orc.OraclePhi3V(cfg, f32(w))
    ids = hf.text_cases(kind)
    S = ids.shape[1]
    got, _ = o(input_ids=ids, max_tokens=1)
    assert got.dtype == torch.float32
    assert close(got[0, -1], GOLD[f"{kind}_last_logits"]) and close(got[0, ::7, ::97], GOLD[f"{kind}_mid_logits_sub"])
    transformers = pytest.importorskip("transformers")
    m = hf.hf_phi3(cfg, w)
    with torch.no_grad():
        ref = m(input_ids=torch.as_tensor(ids)).logits
        assert close(got, ref)
        # KV-cache decode (oracle: prefill S-3 then three single-token calls) == HF full forward
        lg, cache = o(input_ids=ids[:, :S - 3], max_tokens=3)
        for t in range(S - 3, S):
            lg, cache = o(input_ids=ids[:, t:t + 1], cache=cache)
            assert close(lg[0, -1], ref[0, t])
        if kind == "tiny":
            # left-padded batch: `_tokenize` conventions (pad id 0, position id 1, mask 0) vs HF attention_mask + position_ids
            lens = [S, 17, 33]
            width = S
            b_ids = np.stack([np.concatenate([np.zeros(width - n, np.int64), ids[0, :n]]) for n in lens])
            mask = np.stack([np.concatenate([np.zeros(width - n, np.int64), np.ones(n, np.int64)]) for n in lens])
            pids = np.stack([np.concatenate([np.ones(width - n, np.int64), np.arange(n)]) for n in lens])
            got_b, _ = o(input_ids=b_ids, pids=pids, mask=mask, max_tokens=1)
            ref_b = m(input_ids=torch.as_tensor(b_ids), attention_mask=torch.as_tensor(mask), position_ids=torch.as_tensor(pids)).logits
            for r, n in enumerate(lens):                         # valid rows only (pad query rows are Q7: undefined)
                assert close(got_b[r, width - n:], ref_b[r, width - n:]), r


def test_rope_tables_match_hf_longrope():
    """cos/sin of the oracle's SuRoPE == HF's longrope rotary embedding (short and long factors, x 1.19024)."""
    pytest.importorskip("transformers")
    for kind, S in (("tiny", 50), ("tiny_long", 50)):
        cfg = hf.text_config(kind)
        m = hf.hf_phi3(cfg, synth_weights(cfg, seed=0))
        pos = torch.arange(S)[None]
        cos_hf, sin_hf = m.model.rotary_emb(torch.zeros(1, dtype=torch.float32), pos)
        cos, sin = orc.su_rope_tables(cfg, S, None)
        assert torch.allclose(cos[0, 0], cos_hf[0], atol=1e-5) and torch.allclose(sin[0, 0], sin_hf[0], atol=1e-5)


def test_clip_tower_matches_hf_clip_vision_model():
    """oracle.clip_model (23-of-24-layers rule, CLS dropped, no post-LN) == HF CLIPVisionModel hidden_states[-2][:, 1:]."""
    cfg = make_config(tiny_config_dict(vision=True))
    w = synth_weights(cfg, seed=0, std_scale=4.0)
    o = orc.OraclePhi3V(cfg, f32(w))
    pix = hf.clip_pixels(2)
    got = o.clip_model(pix)
    assert close(got[:, ::5], GOLD["clip_tiny_feats_sub"], 1e-3)
    pytest.importorskip("transformers")
    with torch.no_grad():
        ref = hf.hf_clip(cfg, w)(pixel_values=pix, output_hidden_states=True).hidden_states[-2][:, 1:]
    assert close(got, ref, 1e-3)


def test_clip_full_width_two_layers_matches_hf():
    """CLIP-L width (1024, 16 heads, 4096 MLP, 577 tokens), 3 layers of which 2 run."""
    pytest.importorskip("transformers")
    d = tiny_config_dict(vision=True)
    from phi_3_vision_mlx_amd.config import CLIP_L_336
    d["clip"] = dict(CLIP_L_336, num_hidden_layers=3)
    d["img_processor"] = dict(image_dim_out=1024, name="clip_vision_model", num_img_tokens=144)
    cfg = make_config(d)
    w = synth_weights(cfg, seed=0)
    o = orc.OraclePhi3V(cfg, f32(w))
    pix = hf.clip_pixels(1, seed=4)
    with torch.no_grad():
        ref = hf.hf_clip(cfg, w)(pixel_values=pix, output_hidden_states=True).hidden_states[-2][:, 1:]
    assert close(o.clip_model(pix), ref, 1e-3)
