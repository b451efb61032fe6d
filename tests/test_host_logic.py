"""Host-side logic of the drop-in API (no GPU): template, streamer, stoppers, tokenizer,
synthetic weights, config, loader errors -- checked against the reference's documented behaviour."""
import io
import json
import os

import numpy as np
import pytest
import torch


def test_chat_template_matches_reference_format(capsys):
    from phi_3_vision_mlx_amd.api import _apply_chat_template
    p, im = _apply_chat_template("  What is this? ", None, verbose=False)
    assert p == "<|user|>\nWhat is this?<|end|>\n<|assistant|>\n" and im is None
    ps, _ = _apply_chat_template(["a", "b "], None, verbose=False)
    assert ps == ["<|user|>\na<|end|>\n<|assistant|>\n", "<|user|>\nb<|end|>\n<|assistant|>\n"]
    from PIL import Image
    img = Image.new("RGB", (8, 8))
    p, ims = _apply_chat_template("hi", [img, img], verbose=False)
    assert p == "<|user|>\n<|image_1|>\n<|image_2|>\nhi<|end|>\n<|assistant|>\n" and len(ims) == 2
    raw, _ = _apply_chat_template("raw", None, verbose=False, apply_chat_template=False)
    assert raw == "raw"
    with pytest.raises(ValueError):
        _apply_chat_template("x", "/no/such/file.png", verbose=False)


def test_byte_tokenizer_contract():
    from phi_3_vision_mlx_amd.processor import ByteTokenizer
    t = ByteTokenizer()
    ids = t("<|user|>\nhi<|end|>").input_ids
    assert ids[0] == 1 and 32010 in ids and ids[-1] == 32007
    assert t.encode(" The", add_special_tokens=False)[0] == ByteTokenizer.PREFIX      # [1:] drops it (ref :538)
    assert t.decode(t("héllo <|end|>").input_ids) == "héllo <|end|>"
    assert t.batch_decode([[1, 3 + ord("a")], [0, 0, 3 + ord("b")]]) == ["a", "b"]


class _P:
    from phi_3_vision_mlx_amd.processor import ByteTokenizer
    tokenizer = ByteTokenizer()


def test_streamer_batch_trims_at_first_eos_inclusive():
    from phi_3_vision_mlx_amd.api import ID_EOS, Streamer
    s = Streamer(_P, stream=True, mute=False)
    for step in ([[70], [71]], [[ID_EOS], [72]], [[73], [ID_EOS]]):
        s(torch.tensor(step))
    txt, n = s.end()
    assert n == 6 and len(txt) == 2
    assert txt[0].endswith("<|end|>") and "<73>" not in txt[0] and txt[1].endswith("<|end|>")


def test_token_stopper_waits_for_all_rows():
    from phi_3_vision_mlx_amd.api import ID_EOS, TokenStopper
    ts = TokenStopper(_P, 2)
    assert not ts(torch.tensor([[5], [6]]))
    assert not ts(torch.tensor([[ID_EOS], [6]]))
    assert not ts(torch.tensor([[7], [8]]))               # row 0 already finished, row 1 not yet
    assert ts(torch.tensor([[9], [ID_EOS]]))


def test_logit_stopper_only_active_for_int_below_max_tokens():
    from phi_3_vision_mlx_amd.api import LogitStopper
    assert LogitStopper(100, False).early_stop is False
    assert LogitStopper(100, True).early_stop is True           # isinstance(True, int): the reference enables it (as the int 1)
    assert LogitStopper(100, 200).early_stop is False
    assert LogitStopper(100, 20).early_stop == 20


def test_synthetic_weights_deterministic_and_complete():
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.weights import synth_values, synth_weights, weight_specs
    cfg = make_config(tiny_config_dict())
    w1, w2 = synth_weights(cfg, seed=3), synth_weights(cfg, seed=3)
    assert set(w1) == {n for n, _, _ in weight_specs(cfg)}
    assert all(torch.equal(w1[k], w2[k]) for k in w1)
    assert not torch.equal(w1["lm_head.weight"], synth_weights(cfg, seed=4)["lm_head.weight"])
    v = synth_values(200000, 7, 0.02).float()
    assert abs(v.std().item() - 0.02) < 5e-4 and abs(v.mean().item()) < 2e-4
    # chunking must not change the stream (the GPU path uses bigger chunks)
    assert torch.equal(synth_values(5000, 1, 0.02, chunk=64), synth_values(5000, 1, 0.02, chunk=4096))
    full = make_config()
    n_dec = sum(int(np.prod(s)) for n, s, _ in weight_specs(full) if n.startswith("model.layers.") or n == "lm_head.weight")
    assert abs(n_dec * 2 / 1e9 - 7.445) < 0.01                   # 7.445 GB streamed per decoded token


def test_safetensors_roundtrip_and_loader_errors(tmp_path):
    from phi_3_vision_mlx_amd.config import load_config, make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.weights import load_safetensors_dir, save_safetensors_dir, synth_weights
    d = tiny_config_dict(vision=False)
    cfg = make_config(d)
    w = synth_weights(cfg, seed=1)
    save_safetensors_dir(w, d, str(tmp_path / "m"))
    cfg2 = load_config(str(tmp_path / "m" / "config.json"), use_quantized_cache=False)
    back = load_safetensors_dir(str(tmp_path / "m"), cfg2)
    assert all(torch.equal(back[k], w[k]) for k in w)
    with pytest.raises(FileNotFoundError):
        load_config(str(tmp_path / "nope.json"))
    (tmp_path / "bad.json").write_text("{not json")
    with pytest.raises(ValueError):
        load_config(str(tmp_path / "bad.json"))
    del w["lm_head.weight"]
    save_safetensors_dir(w, d, str(tmp_path / "m2"))
    with pytest.raises(KeyError):
        load_safetensors_dir(str(tmp_path / "m2"), cfg2)


def test_load_error_behaviour():
    """Missing model / adapter directories raise FileNotFoundError (the reference fails in _get_cfg / load_weights);
    use_adapter=True resolves adapters/<model dir name> as phi_3_vision_mlx.py:462-464,1316-1317."""
    from phi_3_vision_mlx_amd import api
    with pytest.raises(FileNotFoundError):
        api.load(model_path="/definitely/not/here")
    with pytest.raises(FileNotFoundError):
        api.load(use_adapter=True, model_path="/definitely/not/here")
    assert api._get_adapter_path("models/phi3_v") == "adapters/phi3_v"


def test_shim_exports_reference_api():
    import inspect
    import phi_3_vision_mlx_amd as m
    ref = {"generate": ["prompt", "images", "preload", "blind_model", "quantize_model", "quantize_cache", "use_adapter",
                        "max_tokens", "verbose", "return_tps", "early_stop", "stream", "apply_chat_template", "enable_api"],
           "choose": ["prompt", "choices", "images", "preload", "blind_model", "quantize_model", "quantize_cache",
                      "use_adapter", "verbose", "apply_chat_template"],
           "constrain": ["prompt", "constraints", "images", "preload", "blind_model", "quantize_model", "quantize_cache",
                         "use_adapter", "verbose", "apply_chat_template", "use_beam"],
           "load": ["blind_model", "quantize_model", "quantize_cache", "use_adapter", "kwargs"]}
    for fn, params in ref.items():
        assert list(inspect.signature(getattr(m, fn)).parameters) == params, fn
    bp = inspect.signature(m.benchmark).parameters                      # reference: benchmark(blind_model=False, json_path='benchmark.json')
    assert list(bp)[:2] == ["blind_model", "json_path"] and bp["json_path"].default == "benchmark.json"
    assert all(p.kind is inspect.Parameter.KEYWORD_ONLY for n, p in bp.items() if n not in ("blind_model", "json_path"))
    from phi_3_vision_mlx_amd.api import BENCHMARK_BATCH
    assert len(BENCHMARK_BATCH) == 15                                    # 16 literals, two fused by the reference's missing comma
    assert inspect.signature(m.generate).parameters["max_tokens"].default == 512
    assert inspect.signature(m.constrain).parameters["constraints"].default == [(0, "\nThe"), (100, " The correct answer is"), "ABCDE"]
    assert m.ID_EOS == 32007 and m.ID_ASS == 32001


def test_adapter_files_round_trip_and_layer_selection(tmp_path):
    """adapter_config.json / adapters.safetensors in the reference's format (phi.py:56,61; phi_3_vision_mlx.py:234-245,
    1005-1012): int lora_layers = the LAST n layers, list = indices, scale = scale * alpha / rank, missing tensors = identity."""
    import torch
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.weights import load_adapter, resolve_adapter, save_adapter
    cfg = make_config(tiny_config_dict(vision=False))
    H, nl = cfg.hidden_size, cfg.num_hidden_layers
    t = {f"model.layers.{i}.self_attn.qkv_proj.lora_a": torch.randn(H, 2) for i in range(nl)}
    t.update({f"model.layers.{i}.self_attn.qkv_proj.lora_b": torch.randn(2, 3 * H) for i in range(nl)})
    lora_cfg = {"model_path": "m", "adapter_path": "a", "lora_layers": 1, "lora_targets": ["self_attn.qkv_proj", "mlp.down_proj"],
                "lora_parameters": {"rank": 2, "alpha": 4, "dropout": 0.0, "scale": 10.0}}
    save_adapter(str(tmp_path / "ad"), lora_cfg, t)
    got_cfg, got_t = load_adapter(str(tmp_path / "ad"))
    assert got_cfg == lora_cfg and set(got_t) == set(t)
    ad = resolve_adapter(cfg, got_cfg, got_t)
    assert list(ad) == [f"model.layers.{nl - 1}.self_attn.qkv_proj.weight"]      # last layer only; down_proj has no tensors
    a, b, scale = ad[f"model.layers.{nl - 1}.self_attn.qkv_proj.weight"]
    assert scale == 20.0 and a.dtype == torch.float32 and torch.equal(a, t[f"model.layers.{nl - 1}.self_attn.qkv_proj.lora_a"])
    got_cfg["lora_layers"] = [0]
    assert list(resolve_adapter(cfg, got_cfg, got_t)) == ["model.layers.0.self_attn.qkv_proj.weight"]
    got_cfg["lora_layers"] = "all"
    with pytest.raises(ValueError):
        resolve_adapter(cfg, got_cfg, got_t)
    with pytest.raises(FileNotFoundError):
        load_adapter(str(tmp_path / "missing"))


def test_oracle_lora_is_the_reference_formula():
    """OraclePhi3V.proj == LoRALinear.__call__ (phi.py:129-133) and reduces to nn.Linear when lora_b = 0 (its init)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import torch
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    cfg = make_config(tiny_config_dict(vision=False))
    g = torch.Generator().manual_seed(0)
    w = {"w": (torch.randn(24, 16, generator=g) * 0.3).to(torch.bfloat16)}
    x = torch.randn(3, 16, generator=g).to(torch.bfloat16)
    a, b = torch.randn(16, 4, generator=g), torch.randn(4, 24, generator=g)
    o = orc.OraclePhi3V(cfg, w, adapters={"w": (a, b, 0.5)})
    y = (x.float() @ w["w"].float().t()).to(torch.bfloat16)
    want = (y.float() + 0.5 * ((x.float() @ a) @ b)).to(torch.bfloat16)
    assert torch.equal(o.proj(x, "w"), want)
    o0 = orc.OraclePhi3V(cfg, w, adapters={"w": (a, torch.zeros_like(b), 0.5)})
    assert torch.equal(o0.proj(x, "w"), y)


def test_mlx_int4_format_round_trip_and_checkpoint_loading(tmp_path):
    """MLX nn.quantize format (phi_3_vision_mlx.py:297-305): 8 codes per uint32 (code k at bits 4k), group-64 scales/biases;
    w ~ scale*q+bias within half a step; a quantised checkpoint directory loads with decoder projections kept 4-bit
    (Q4Weight) and everything else dequantised; the device repacking is a pure permutation of the codes."""
    import torch
    from safetensors.torch import save_file
    from phi_3_vision_mlx_amd.config import load_config, make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.weights import (Q4Weight, load_safetensors_dir, mlx_dequantize, mlx_quantize, mlx_unpack, q4_repack,
                                              synth_weights)
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(8, 192, generator=g) * 0.05).to(torch.bfloat16)
    packed, sc, bi = mlx_quantize(w)
    assert packed.shape == (8, 24) and sc.shape == (8, 3) and packed.dtype == torch.int32
    q = mlx_unpack(packed)
    assert q.min() >= 0 and q.max() <= 15
    assert int(q[0, 0]) == int(packed[0, 0].item() & 15) and int(q[0, 1]) == int((packed[0, 0].item() >> 4) & 15)
    d = mlx_dequantize(packed, sc, bi)
    step = (w.float().reshape(8, 3, 64).amax(-1) - w.float().reshape(8, 3, 64).amin(-1)) / 15
    assert ((d - w.float()).abs().reshape(8, 3, 64) <= 1.0 * step[..., None] + 2e-3).all()       # the exact-edge rescaling can clip the far end by almost a step
    w4, sb = q4_repack(packed, sc, bi)
    pos = [0, 16, 4, 20, 8, 24, 12, 28]
    back = torch.stack([(w4.to(torch.int64) >> p) & 15 for p in pos], dim=-1).reshape(8, 192)
    assert torch.equal(back, q)
    assert torch.equal((sb.to(torch.int64) & 0xFFFF).to(torch.int16).view(torch.bfloat16), sc)
    # a quantised checkpoint directory
    dcfg = tiny_config_dict(vision=False)
    cfg = make_config(dcfg)
    ws = synth_weights(cfg, seed=2)
    tensors = {}
    for k, v in ws.items():
        if v.dim() == 2 and v.shape[1] % 64 == 0 and ("proj" in k or k in ("lm_head.weight", "model.embed_tokens.weight")):
            p_, s_, b_ = mlx_quantize(v)
            tensors[k], tensors[k[:-7] + ".scales"], tensors[k[:-7] + ".biases"] = p_, s_, b_
        else:
            tensors[k] = v
    (tmp_path / "q").mkdir()
    save_file(tensors, str(tmp_path / "q" / "quantized_model.safetensors"))
    (tmp_path / "q" / "config.json").write_text(json.dumps(dict(dcfg, quantized={"group_size": 64, "bits": 4})))
    cfg_q = load_config(str(tmp_path / "q" / "config.json"))
    got = load_safetensors_dir(str(tmp_path / "q"), cfg_q)
    assert isinstance(got["model.layers.0.self_attn.qkv_proj.weight"], Q4Weight) and isinstance(got["lm_head.weight"], Q4Weight)
    emb = got["model.embed_tokens.weight"]
    # (the embedding table is an mx.dequantize ARRAY in the reference -- nn.QuantizedEmbedding -- multiply and add each rounding to bf16)
    assert emb.dtype == torch.bfloat16 and torch.equal(emb, mlx_dequantize(*mlx_quantize(ws["model.embed_tokens.weight"]), dtype=torch.bfloat16))
    assert torch.equal(got["model.norm.weight"], ws["model.norm.weight"])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("group", [32, 64])
def test_two_independent_statements_of_mx_quantize_agree_bit_for_bit(dtype, group):
    """The MLX stand-in the reference runs over (tests/golden/mlx_shim.py: mx.quantize / mx.dequantize as mlx 0.15.0's composites,
    written over the stand-in's own array primitives, importing nothing of the product) and the product's quantiser
    (weights.mlx_quantize / mlx_dequantize, straight torch) are two separately written statements of one algorithm: codes, scales
    and biases must agree BIT FOR BIT on random inputs of every dtype -- both branches of the edge selection, all-zero, constant,
    one-sided, denormal-range and huge groups included -- and so must the dequantised arrays.  (Until round 5 the stand-in
    called the product's function: the reference-composed q4 fixtures compared the quantiser with itself.)"""
    import torch
    import mlx_shim as mx
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    assert "phi_3_vision_mlx_amd" not in open(mx.__file__).read().replace("`phi_3_vision_mlx_amd.weights.mlx_quantize`", "")
    dt = getattr(torch, dtype)
    as_bits = (lambda t: t.view(torch.int32)) if dt == torch.float32 else (lambda t: t.view(torch.int16))
    gen = torch.Generator().manual_seed(group)
    sides = set()
    for trial in range(24):
        std = (0.02, 3.0, 1e-4, 300.0)[trial % 4]
        w = (torch.randn(48, 4 * group, generator=gen) * std).to(dt)
        if trial % 3 == 0:
            w[0] = 0                                              # all-zero groups: the scale clamps to 1e-7 (in w's dtype)
            w[1] = 0.37                                           # constant groups
            w[2, :group] = -w[2, :group].abs() - 0.01             # one-sided negative: the minimum is the exact edge
            w[3, :group] = w[3, :group].abs() + 0.01              # one-sided positive
            w[4, :group] = 0
            w[4, 5] = 1e-9                                        # a range below the clamp
            w[5, :group] = torch.linspace(-1, 1, group).to(dt)    # symmetric: |min| == |max| takes the maximum's branch
        a_w, a_s, a_b = mx.quantize(mx.array(w), group, 4)
        b_w, b_s, b_b = mlx_quantize(w, group, 4)
        assert a_s._t.dtype == dt and b_s.dtype == dt
        assert torch.equal(a_w._t, b_w), f"trial {trial}: packed codes differ"
        assert torch.equal(as_bits(a_s._t), as_bits(b_s)) and torch.equal(as_bits(a_b._t), as_bits(b_b)), f"trial {trial}: scales / biases differ"
        d_a = mx.dequantize(a_w, a_s, a_b, group, 4)._t
        assert d_a.dtype == dt and torch.equal(as_bits(d_a), as_bits(mlx_dequantize(b_w, b_s, b_b, group, 4, dtype=dt)))
        sides |= set((b_s.float() > 0).unique().tolist())
        g_ = w.float().reshape(48, 4, group)
        step = (g_.amax(-1) - g_.amin(-1)) / 15
        slack = 2.0 ** -7 * g_.abs().amax(-1) if dt != torch.float32 else 0.0    # narrow dtypes: each primitive's own rounding
        assert ((d_a.float().reshape(48, 4, group) - g_).abs() <= (1.01 * step + 1e-6 + slack)[..., None]).all(), "not within a step"
    assert sides == {False, True}                                 # both signs of the scale were exercised


@pytest.mark.parametrize("name", ["mlx_quantize_doc_example.json", "mlx_quantize_doc_example2.json"])
def test_mlx_quantize_documented_example(name):
    """tests/golden/mlx_quantize_doc_example*.json: hand-worked instances of the affine group quantisation mx.quantize
    documents (formula, 8 four-bit codes per uint32 with element k in bits [4k, 4k+4), the larger-magnitude end of the range
    represented exactly).  Example 1: positive scale, bias = minimum; example 2 (round 5, derived independently of the loader's
    code): NEGATIVE scale, bias = maximum, scrambled code order.  They pin what the loader ASSUMES about
    `quantized_model.safetensors` (phi_3_vision_mlx.py:297-305); no MLX-written file exists in this environment to pin it harder."""
    import json
    import os
    import torch
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize, mlx_unpack
    ex = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))
    w = torch.tensor(ex["weights"], dtype=torch.float32)
    packed = torch.tensor([[int(h, 16) for h in row] for row in ex["packed_hex"]], dtype=torch.int64)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
    scales, biases = torch.tensor(ex["scales"]), torch.tensor(ex["biases"])
    assert mlx_unpack(packed).tolist() == ex["codes"]
    assert torch.equal(mlx_dequantize(packed, scales, biases), w)          # s * q + beta reproduces the ramp exactly
    p2, s2, b2 = mlx_quantize(w)
    assert torch.equal(p2, packed) and torch.equal(s2.float(), scales) and torch.equal(b2.float(), biases)
