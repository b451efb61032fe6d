"""The oracle against the REFERENCE'S OWN model / loop code (CPU).

`tests/golden/ref_model_tiny.npz` was written by `tests/golden/gen_golden_refmodel.py`, which executes /root/reference's
`phi.py` model classes and `phi_3_vision_mlx.py` loops over the functional MLX stand-in `tests/golden/mlx_shim.py` on the
tiny synthetic checkpoint.  These tests pin `oracle/phi3v_oracle.py` (and with it every `*_oracle.npz` fixture) to those
outputs:

  * text paths (B = 1, left-padded batch, > 4096-token long-factor prompt, LoRA adapter): logits BIT-EXACT, tokens exact;
  * image paths (336x336, 640x480, two images in one prompt): tokens exact, logits within 2^-7 of the row's max|logit| (<= 2 bf16 ulps of it) -- the CLIP tower
    runs on fp32 activations (phi.py:279, 309) and the two programs' fp32 matmuls associate differently (2e-6 relative),
    which the fp32 -> bf16 scatter (phi.py:414) turns into an occasional last-bit flip;
  * `_choose_from` / `_constrain` (plain and beam): every model call of the reference's loop is reproduced -- the ids fed,
    (advance_offset, n_beam), and per position the top-8 logits, the log-sum-exp and the constraint-id logits.

When /root/reference is present (build container) `test_live_*` additionally run the reference side by side with the
oracle on inputs that are NOT in the fixture.  tests/test_model_gpu.py checks the HIP path against the same fixture.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)
import phi3v_oracle as orc  # noqa: E402
from golden_inputs import make_image  # noqa: E402
from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict  # noqa: E402
from phi_3_vision_mlx_amd.processor import Phi3FProcessor, Phi3VProcessor  # noqa: E402
from phi_3_vision_mlx_amd.weights import peaked_lm_head, synth_weights  # noqa: E402

IMAGES = {"sq": (336, 336, "noise", 0), "land": (640, 480, "smooth", 1)}
TINY_PROMPTS = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
VIS_PROMPT = "<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n"
VIS2_PROMPT = "<|user|>\n<|image_1|>\n<|image_2|>\nCompare the two.<|end|>\n<|assistant|>\n"
LONG_PROMPT = "<|user|>\n" + ("the quick brown fox jumps over the lazy dog. " * 92) + "<|end|>\n<|assistant|>\n"
CASES = {"text": (True, TINY_PROMPTS[0], None), "batch": (True, TINY_PROMPTS, None), "long": (True, LONG_PROMPT, None),
         "vis": (False, VIS_PROMPT, ["sq"]), "visns": (False, VIS_PROMPT, ["land"]), "vis2": (False, VIS2_PROMPT, ["sq", "land"]),
         "lora": (True, TINY_PROMPTS[1], None)}


def _bits(t):
    return t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def _from_bits(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(torch.bfloat16)


@pytest.fixture(scope="module")
def fx():
    g = np.load(os.path.join(GOLDEN, "ref_model_tiny.npz"))
    with open(os.path.join(GOLDEN, "ref_model_tiny.json")) as f:
        meta = json.load(f)
    return g, meta


class Tiny:
    def __init__(self, blind):
        self.cfg = make_config(tiny_config_dict(vision=not blind))
        self.w = synth_weights(self.cfg, seed=0, std_scale=4.0)
        self.base = self.w["lm_head.weight"]
        self.oracle = orc.OraclePhi3V(self.cfg, dict(self.w), cache_fp32=True)
        self.proc = (Phi3FProcessor if blind else Phi3VProcessor)(None)

    def head(self, spread, hs):
        self.oracle.w["lm_head.weight"] = peaked_lm_head(self.base, float(spread), int(hs))
        self.oracle._f32.pop("lm_head.weight", None)


@pytest.fixture(scope="module")
def tiny_models():
    return {True: Tiny(True), False: Tiny(False)}


def _adapter_from_fixture(g, meta, cfg):
    from phi_3_vision_mlx_amd.weights import resolve_adapter
    tensors = {k[len("lora_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("lora_model")}
    return resolve_adapter(cfg, meta["lora_adapter"], tensors)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_generate_matches_reference(fx, tiny_models, name):
    g, meta = fx
    blind, prompt, images = CASES[name]
    t = tiny_models[blind]
    t.head(g["spread"][0], g[name + "_head_seed"][0])
    t.oracle.adapters = _adapter_from_fixture(g, meta, t.cfg) if name == "lora" else {}
    try:
        imgs = [make_image(*IMAGES[i]) for i in images] if images else None
        inp = t.proc(prompt, imgs) if imgs else t.proc(prompt)
        # the build's processor reproduces the reference processor's model inputs
        assert np.array_equal(np.asarray(inp["input_ids"]), g[name + "_input_ids"])
        if name + "_pids" in g.files:
            assert np.array_equal(np.asarray(inp["pids"]), g[name + "_pids"]) and np.array_equal(np.asarray(inp["mask"]), g[name + "_mask"])
        if imgs:
            import hashlib
            pv = np.ascontiguousarray(np.asarray(inp["pixel_values"], dtype=np.float64).astype(np.float32))
            assert hashlib.sha256(pv.tobytes()).hexdigest() == meta[name]["pixel_values_f32_sha256"]
            assert np.asarray(inp["image_sizes"]).tolist() == meta[name]["image_sizes"]
            assert np.asarray(inp["positions"]).shape[0] == meta[name]["n_positions"]
        n = g[name + "_tokens"].shape[1]
        o_in = {k: (torch.from_numpy(np.asarray(v)) if k == "pixel_values" else v) for k, v in inp.items()}
        toks, lgs = orc.greedy_generate(t.oracle, o_in, n, stop_on_eos=False)
        assert np.array_equal(toks.numpy(), g[name + "_tokens"]), f"{name}: tokens differ from the reference's"
        ref = _from_bits(g[name + "_logits_bf16"])
        if images is None:
            assert np.array_equal(_bits(lgs), g[name + "_logits_bf16"]), f"{name}: text-path logits are not bit-exact"
        else:
            ulp = 2.0 ** -7 * ref.float().abs().amax(-1, keepdim=True)         # <= 2 bf16 ulps of the row's largest logit
            worst = ((lgs.float() - ref.float()).abs() / ulp).max().item()
            assert worst <= 1.0, f"{name}: image-path logits off by {worst:.2f} x 2^-7 max|logit|"
        assert (g[name + "_margins"] > 1.0).all()
    finally:
        t.oracle.adapters = {}


def test_oracle_quantised_cache_is_the_references(fx):
    """Round 5: the reference's own quantised KV cache (`use_quantized_cache`, phi.py:528-540: mx.quantize(group 32, 4 bits) of the
    first call's keys / values, later tokens unquantised) restated in the oracle (OracleKVCache + mx_quantize) reproduces the
    reference's `_generate` run of fixture case `q4cache` BIT FOR BIT: every logit of every step."""
    g, meta = fx
    cfg = make_config(tiny_config_dict(vision=False), use_quantized_cache=True, cache_format="mlx4")
    w = synth_weights(cfg, seed=0, std_scale=4.0)
    w["lm_head.weight"] = peaked_lm_head(w["lm_head.weight"], float(g["spread"][0]), int(g["q4cache_head_seed"][0]))
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    n = g["q4cache_tokens"].shape[1]
    toks, lgs = orc.greedy_generate(o, {"input_ids": g["q4cache_input_ids"]}, n, stop_on_eos=False)
    assert np.array_equal(toks.numpy(), g["q4cache_tokens"])
    assert np.array_equal(_bits(lgs), g["q4cache_logits_bf16"]), "quantised-cache logits are not bit-exact"
    assert (g["q4cache_margins"] > 1.0).all() and float(g["q4cache_rel_tol"][0]) == pytest.approx(0.10)


def test_clear_constrain_heads_and_well_conditioned_fixture(fx, tiny_models):
    """Round 5 fixtures.  (a) `loops_clear`: under the recorded head seeds EVERY decision of the oracle's constrain loop (plain and
    beam) has a margin above 3 % of max |logit|, and the oracle's synthesised ids decode to the reference's final texts.  (b)
    ref_model_wc: 128 reference steps per head at rel_tol <= 1.5 %, >= 100 clear under the unsearched peaked head, >= 85 under the
    plain head."""
    g, meta = fx
    t = tiny_models[True]
    lc = meta["loops_clear"]
    cons = tuple(lc["constraint"])
    idc = t.proc.tokenizer.encode(cons[1], add_special_tokens=False)[1:]
    for ub in (False, True):
        info = lc[f"beam{int(ub)}"]
        t.head(g["spread"][0], info["head_seed"])
        ps = TINY_PROMPTS if not ub else TINY_PROMPTS[:1] * 2
        tr = []
        synth, _ = orc.constrain_one(t.oracle, dict(t.proc(list(ps))), cons, idc, use_beam=ub, trace=tr)
        assert min(m for _, m, _ in tr) > lc["clear_margin"] and len(tr) == info["n_decisions"]
        assert np.array_equal(synth.numpy(), g[f"clear_beam{int(ub)}_synth"])
        assert all(txt.endswith(cons[1]) for txt in info["full_text"])
    wc = np.load(os.path.join(GOLDEN, "ref_model_wc.npz"))
    assert float(wc["rel_tol"][0]) <= 0.015 + 1e-9 and wc["plain_tokens"].shape[1] == 128 and wc["peaked0_tokens"].shape[1] == 128
    assert int((wc["peaked0_margins"] > 1.0).sum()) >= 100 and int((wc["plain_margins"] > 1.0).sum()) >= 85


def test_round6_well_conditioned_fixtures_are_what_the_gpu_tests_assume():
    """ref_model_wc_c2 (the reference's own run of BASELINE config 2's image request on the well-conditioned checkpoint) and the two
    config-5 fixtures of the same checkpoint (oracle + the build's quantisers): tolerances at or below what VERDICT r05 item 1c asked
    for (2 % / 8 %), the recorded step counts, and enough CLEAR steps under the unsearched heads for the token assertions of
    tests/test_model_gpu.py to mean something."""
    import json
    c2 = np.load(os.path.join(GOLDEN, "ref_model_wc_c2.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "ref_model_wc_c2.json")))
    assert meta["generator"].endswith("wc_c2") and float(c2["rel_tol"][0]) <= 0.02 + 1e-9 and int(c2["n_ids"][0]) == 2531
    assert c2["plain_tokens"].shape[1] == 16 and c2["peaked0_tokens"].shape[1] == 16
    assert int((c2["plain_margins"] > 1.0).sum()) + int((c2["peaked0_margins"] > 1.0).sum()) >= 16
    for name, cap in (("c5_wc", 0.08), ("c5w_wc", 0.03)):
        g5 = np.load(os.path.join(GOLDEN, f"{name}_oracle.npz"))
        assert float(g5["rel_tol"][0]) <= cap + 1e-9 and int(g5["n_ids"][0]) == 2531 and abs(float(g5["residual_scale"][0]) - 1 / 1024) < 1e-9
        assert g5["plain_tokens"].shape[1] == 8 and g5["peaked0_tokens"].shape[1] == 8
        assert int((g5["plain_margins"] > 1.0).sum()) + int((g5["peaked0_margins"] > 1.0).sum()) >= 3
    q4 = np.load(os.path.join(GOLDEN, "q4_wc_oracle.npz"))          # MLX 4-bit weights, config 1's prompt, 16 steps
    assert float(q4["rel_tol"][0]) <= 0.02 + 1e-9 and int(q4["n_ids"][0]) == 128 and abs(float(q4["residual_scale"][0]) - 1 / 1024) < 1e-9
    assert q4["plain_tokens"].shape[1] == 16 and np.array_equal(q4["ids"], np.load(os.path.join(GOLDEN, "c1_oracle.npz"))["ids"])
    assert int((q4["plain_margins"] > 1.0).sum()) >= 8 and int((q4["peaked0_margins"] > 1.0).sum()) >= 8


def test_full_size_oracle_fixtures_equal_the_reference_composed_ones():
    """FULL SIZE (32 layers x 3072, vocab 32064; CLIP ViT-L/14-336 on 17 crops): `ref_model_full.npz` holds what the REFERENCE'S
    own `_load` + processors + `_generate` + `Phi3VForCausalLM` produce (over the MLX stand-in) for the requests of the oracle
    fixtures c1 (128 ids, text) and c2 (bench.py's image request, 2531 ids) under those fixtures' lm_heads.  The oracle fixtures,
    generated independently by `gen_golden_oracle.py`, must agree: tokens exactly; logits bit for bit on the text path; on the
    image path within the spread of two correct programs at this depth (measured 1.5-2.4 % of max|logit|)."""
    g = np.load(os.path.join(GOLDEN, "ref_model_full.npz"))
    for name, exact in (("c1", True), ("c2", False)):
        o = np.load(os.path.join(GOLDEN, f"{name}_oracle.npz"))
        assert int(o["head_seed"][0]) == int(g[name + "_head_seed"][0])
        n = min(o["tokens"].shape[1], g[name + "_tokens"].shape[1])
        assert n >= 3 and np.array_equal(o["tokens"][:, :n], g[name + "_tokens"][:, :n]), name
        a, b = _from_bits(o["logits_bf16"][:, :n]).float(), _from_bits(g[name + "_logits_bf16"][:, :n]).float()
        if exact:
            # bit for bit, but for entries that cancel to ~0 in the PREFILL row (the oracle fixture projects the last position only,
            # the reference all 128: another GEMM blocking -- 7 of 128256 entries, <= 2e-6 absolute)
            assert (a - b).abs().le(2.0 ** -18 * b.abs().max()).all() and (a != b).sum().item() <= 32, f"{name}: full-size text path"
            assert np.array_equal(o["logits_bf16"][:, 1:n], g[name + "_logits_bf16"][:, 1:n]), f"{name}: decode steps not bit-exact"
        else:
            # 23 fp32 CLIP layers whose matmuls associate differently, one bf16 scatter, then 32 decoder layers that amplify any
            # flipped bf16 bit (~3 % of max|logit| at this context): two correct programs, 1.5-2.4 % apart
            worst = ((a - b).abs().amax(-1) / b.abs().amax(-1)).max().item()
            assert worst <= 0.035, f"{name}: {100 * worst:.1f} % of max|logit|"
        assert (g[name + "_margins"][:, :n] > 1.0).all()


@pytest.mark.parametrize("name", ["q4text", "q4vis", "q4batch"])
def test_reference_written_4bit_checkpoint_loader_and_oracle(fx, tmp_path, name):
    """Row f4 against the reference's own writer and reader: `_quantize` (phi_3_vision_mlx.py:291-305) wrote the checkpoint, `_load`
    (`nn.quantize` before `load_weights`, :264) read it back and `_generate` ran on it (fixture).  Here: (1) the same bytes come out
    of the seeded weights + `weights.mlx_quantize` (sha256 of every tensor of the reference-written file); (2) the build's loader
    takes the directory: decoder projections + lm_head stay 4-bit (`Q4Weight`), both embeddings, the CLIP position table, the ViT's
    and the projector's Linears are dequantised, the sanitised patch convolution is permuted back; (3) the oracle on the
    dequantised values (fp32 scale * q + bias, what MLX's quantised matmul accumulates; bf16 for embedding rows) reproduces the
    reference's logits -- bit for bit on the text prompt and on the two prompts as ONE left-padded batch (q4batch, round 6:
    QuantizedLinear at B = 2 under the reference's Mask4D / position ids), to 2^-5 of the row's largest logit on the image prompt
    (fp32 tower)."""
    import q4_ckpt
    from phi_3_vision_mlx_amd.weights import Q4Weight, load_safetensors_dir, mlx_dequantize
    g, meta = fx
    cfg_d, tensors = q4_ckpt.build(str(tmp_path), meta[name], g[name + "_head_seed"][0], float(g["spread"][0]))
    cfg = make_config(cfg_d)
    assert cfg.quantized == {"group_size": 64, "bits": 4} and cfg.sanitized is True
    loaded = load_safetensors_dir(str(tmp_path), cfg)
    q4 = {k for k, v in loaded.items() if isinstance(v, Q4Weight)}
    assert q4 == {k for k in loaded if k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith("_proj.weight"))}
    assert loaded["model.embed_tokens.weight"].dtype == torch.bfloat16
    pe = "model.vision_embed_tokens.img_processor.vision_model.embeddings.patch_embedding.weight"
    assert tuple(loaded[pe].shape) == (128, 3, 14, 14)
    ow = {}
    for k, v in loaded.items():                                   # the oracle's weights: exact dequantised values where MLX multiplies by them
        base = k[:-len(".weight")]
        if isinstance(v, Q4Weight):
            ow[k] = mlx_dequantize(*v)
        elif base + ".scales" in tensors and "embed_tokens" not in k and "position_embedding" not in k:
            ow[k] = mlx_dequantize(tensors[k], tensors[base + ".scales"], tensors[base + ".biases"])
        else:
            ow[k] = v
    oracle = orc.OraclePhi3V(cfg, ow, cache_fp32=True)
    blind, prompt, images = {"q4text": (None, TINY_PROMPTS[1], None), "q4batch": (None, TINY_PROMPTS, None)}.get(name, (None, VIS_PROMPT, ["sq"]))
    proc = Phi3VProcessor(None)
    imgs = [make_image(*IMAGES[i]) for i in images] if images else None
    inp = proc(prompt, imgs) if imgs else proc(prompt)
    n = g[name + "_tokens"].shape[-1]
    o_in = {k: (torch.from_numpy(np.asarray(v)) if k == "pixel_values" else v) for k, v in inp.items()}
    toks, lgs = orc.greedy_generate(oracle, o_in, n, stop_on_eos=False)
    assert np.array_equal(toks.numpy(), g[name + "_tokens"])
    ref = _from_bits(g[name + "_logits_bf16"]).float()
    if images is None:
        assert np.array_equal(_bits(lgs), g[name + "_logits_bf16"]), "4-bit text path: oracle not bit-exact to the reference's QuantizedLinear run"
    else:
        # the fp32 tower's matmuls associate differently in the two programs (1e-7) and the tiny net (std_scale 4) amplifies a flipped
        # bf16 rounding: 2 - 5 bf16 ulps of the largest logit (round 6 head seed: 1.27 x 2^-6 at the worst step; round 5's: 0.9 x 2^-6)
        worst = ((lgs.float() - ref).abs() / (2.0 ** -5 * ref.abs().amax(-1, keepdim=True))).max().item()
        print(f"{name}: oracle vs reference, worst |dlogit| = {worst:.3f} x 2^-5 of the row's largest logit")
        assert worst <= 1.0


class Rec:
    """Records every call of the oracle model the way tests/golden/ref_env.Recorder records the reference's."""

    def __init__(self, model):
        self.model, self.calls = model, []

    def __call__(self, *a, **k):
        logits, cache = self.model(*a, **k)
        self.calls.append(dict(input_ids=torch.as_tensor(np.asarray(k.get("input_ids", a[0] if a else None))).long(), logits=logits,
                               advance_offset=k.get("advance_offset"), n_beam=k.get("n_beam", 1)))
        return logits, cache


def _check_calls(calls, g, prefix, idc):
    ids, args = g[prefix + "ids"], g[prefix + "args"]
    assert len(calls) == ids.shape[0], f"{prefix}: {len(calls)} model calls, the reference made {ids.shape[0]}"
    P = g[prefix + "topv"].shape[2]
    for k, c in enumerate(calls):
        b, l = c["input_ids"].shape
        assert (b, l, -1 if c["advance_offset"] is None else c["advance_offset"], c["n_beam"]) == tuple(args[k]), (prefix, k, args[k])
        assert np.array_equal(c["input_ids"].numpy(), ids[k, :b, :l]), f"{prefix} call {k}: other ids were fed than in the reference"
        lg = c["logits"][:, -P:] if l > P else c["logits"]
        p = lg.shape[1]
        v, i = torch.sort(lg.float(), dim=-1, descending=True, stable=True)
        assert np.array_equal(i[..., :8].numpy(), g[prefix + "topi"][k, :b, :p]), f"{prefix} call {k}: top-8 ids differ"
        assert np.array_equal(_bits(v[..., :8]), g[prefix + "topv"][k, :b, :p]), f"{prefix} call {k}: top-8 logits differ"
        assert np.array_equal(_bits(lg[..., torch.as_tensor(idc).long()]), g[prefix + "cons"][k, :b, :p])
        np.testing.assert_allclose(torch.logsumexp(lg.float(), dim=-1).numpy(), g[prefix + "lse"][k, :b, :p], rtol=0, atol=1e-5)


def test_oracle_choose_and_constrain_match_reference_loops(fx, tiny_models):
    """`_choose_from` and `_constrain` (phi_3_vision_mlx.py:466-487, 500-619): the oracle's restated loops make the same model
    calls with bit-identical outputs, and reach the reference's final texts."""
    from phi_3_vision_mlx_amd.api import _preprocess
    g, meta = fx
    t = tiny_models[True]
    t.head(g["spread"][0], g["loops_head_seed"][0])
    L = meta["loops"]
    opts = t.proc([f" {c}" for c in "ABCDE"])["input_ids"][:, -1]
    assert ["ABCDE"[i] for i in orc.choose_from(t.oracle, t.proc(TINY_PROMPTS), opts)] == L["choose"]
    assert "ABCDE"[orc.choose_from(t.oracle, t.proc(TINY_PROMPTS[1]), opts)[0]] == L["choose_single"]
    cons = tuple(L["constraint"])
    idc = t.proc.tokenizer.encode(cons[1], add_special_tokens=False)[1:]
    for ub in (False, True):
        ps = TINY_PROMPTS if not ub else TINY_PROMPTS[:1] * 2
        inp = t.proc([_preprocess(p) for p in ps])
        rec = Rec(t.oracle)
        synth, score = orc.constrain_one(rec, dict(inp), cons, idc, use_beam=ub)
        _check_calls(rec.calls, g, f"constrain_beam{int(ub)}_", idc)
        # final text: prompt ids + synthesis, cut at the first EOS after the prompt, ids 0/1 dropped (:598-606)
        S = np.asarray(inp["input_ids"]).shape[1]
        rows = torch.cat([torch.as_tensor(np.asarray(inp["input_ids"])).long(), synth], dim=1).tolist()
        rows = [(r[:r.index(orc.ID_EOS, S)] if orc.ID_EOS in r[S:] else r) for r in rows]
        rows = [[x for x in r if x not in (0, 1)] for r in rows]
        texts = [_preprocess(s) for s in t.proc.tokenizer.batch_decode(rows)]
        assert texts == L[f"constrain_beam{int(ub)}"]["full_text"]


# ------------------------------------------------------------------------------------------------------------------------
# live: the reference next to the oracle (build container only)
# ------------------------------------------------------------------------------------------------------------------------
def _live():
    import ref_env
    if not ref_env.available():
        pytest.skip("needs /root/reference (build container only)")
    return ref_env


@pytest.fixture(scope="module")
def live_blind(tmp_path_factory):
    ref_env = _live()
    from phi_3_vision_mlx_amd.processor import ByteTokenizer
    from phi_3_vision_mlx_amd.weights import save_safetensors_dir
    d = tiny_config_dict(vision=False)
    cfg = make_config(d)
    w = synth_weights(cfg, seed=3, std_scale=4.0, lm_head_spread=4.0, lm_head_seed=1)
    path = str(tmp_path_factory.mktemp("refblind"))
    save_safetensors_dir(w, d, path)
    model, proc = ref_env.load_model(path, ByteTokenizer())
    return ref_env, model, proc, orc.OraclePhi3V(cfg, w, cache_fp32=True), Phi3FProcessor(None)


def test_live_all_positions_rewind_and_beam_view(live_blind):
    """Model-level contract (phi.py:576-592, 509-548): all-position logits of a prefill, a cached L > 1 call that does not
    advance the cache (advance_offset=0), one that commits a single token (advance_offset=1), and a read-only beam view
    (n_beam=3) -- reference and oracle bit for bit, including the cache offsets."""
    ref_env, model, proc, oracle, my = live_blind
    mx = ref_env.load_reference()[0]
    prompts = ["<|user|>\nabc<|end|>\n<|assistant|>\n", "<|user|>\na much longer second prompt<|end|>\n<|assistant|>\n"]
    r_in, o_in = proc(prompts), my(prompts)
    rl, rc = model(**r_in, max_tokens=12)
    ol, oc = oracle(**o_in, max_tokens=12)
    valid = torch.as_tensor(np.asarray(o_in["mask"])).bool()
    assert torch.equal(rl._t[valid], ol[valid])                         # pad query rows are undefined in the reference (Q7)
    assert rc[0].offset == oc[0].offset
    step = torch.tensor([[5, 9, 11], [7, 8, 12]])
    for adv in (0, 1, None):
        rl, rc = model(input_ids=mx.array(step.numpy()), cache=rc, advance_offset=adv)
        ol, oc = oracle(input_ids=step, cache=oc, advance_offset=adv)
        assert torch.equal(rl._t, ol) and rc[0].offset == oc[0].offset, adv
    beam = torch.tensor([[3, 4]] * 6)
    rl, _ = model(input_ids=mx.array(beam.numpy()), cache=rc, n_beam=3, advance_offset=0)
    ol, _ = oracle(input_ids=beam, cache=oc, n_beam=3, advance_offset=0)
    assert torch.equal(rl._t, ol) and rc[0].offset == oc[0].offset
    for a, b in zip(rc, oc):                                            # the caches themselves: same fp32 contents
        assert torch.equal(a.kv._t[:, :, :, :a.offset], b.kv[:, :, :, :b.offset])


def test_live_generate_with_eos_and_streamer(live_blind, capsys):
    """`_generate` end to end incl. Streamer / TokenStopper (phi_3_vision_mlx.py:45-117, 376-409) vs the build's host loop
    driven by the ORACLE model: same texts, same stop step when a row emits EOS."""
    ref_env, model, proc, oracle, my = live_blind
    loops = ref_env.load_reference()[2]
    prompts = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nhello<|end|>\n<|assistant|>\n"]
    texts = loops._generate(model, proc, prompts, max_tokens=9, verbose=False, stream=False, mute=True)
    toks, _ = orc.greedy_generate(oracle, dict(my(prompts)), 9)
    rows = [(r[:r.index(orc.ID_EOS) + 1] if orc.ID_EOS in r else r) for r in toks.tolist()]
    assert my.tokenizer.batch_decode(rows) == texts


def test_shim_semantics():
    """The stand-in's documented MLX semantics, spot-checked (promotion, weak scalars, slice assignment, first-max argmax,
    repeat / tile / split, masked softmax, double-rounded RMSNorm, composite log_softmax)."""
    import mlx_shim as mx
    bf, f32 = mx.bfloat16, mx.float32
    a = mx.array([1.0, 2.0, 3.0]).astype(bf)
    assert (a * mx.array([1.0, 1.0, 1.0])).dtype == f32 and (a * 2.5).dtype == bf and (a + 1).dtype == bf
    assert mx.array([1, 2]).dtype == mx.int32 and mx.array(np.zeros(2)).dtype == f32 and mx.arange(3).dtype == mx.int32
    z = mx.zeros((2, 4), dtype=bf)
    z[1, 1:3] = mx.array([[1.5, 2.5]])
    assert z.tolist() == [[0, 0, 0, 0], [0, 1.5, 2.5, 0]] and z.dtype == bf
    assert mx.argmax(mx.array([1.0, 3.0, 3.0]), axis=-1).item() == 1
    assert mx.repeat(mx.array([[1, 2]]), 2, axis=0).tolist() == [[1, 2], [1, 2]]
    assert mx.repeat(mx.array([1, 2]), 2, axis=0).tolist() == [1, 1, 2, 2] and mx.tile(mx.array([1, 2]), (2, 1)).tolist() == [[1, 2], [1, 2]]
    assert [p.tolist() for p in mx.split(mx.arange(6), [2, 5])] == [[0, 1], [2, 3, 4], [5]]
    s = mx.softmax(mx.array([[0.0, -mx.inf], [-mx.inf, -mx.inf]]), axis=-1)
    assert s.tolist() == [[1.0, 0.0], [0.0, 0.0]]
    x = mx.array(np.linspace(-3, 3, 64, dtype=np.float32)).astype(bf)
    w = mx.array(np.linspace(0.9, 1.1, 64, dtype=np.float32)).astype(bf)
    y = mx.fast_rms_norm(x, w, 1e-5)
    xf = x._t.float()
    once = (xf * torch.rsqrt(xf.pow(2).mean() + 1e-5)).to(torch.bfloat16)
    assert torch.equal(y._t, once * w._t) and y.dtype == bf
    lg = mx.array(np.random.default_rng(0).normal(0, 3, (2, 500)).astype(np.float32)).astype(bf)
    ls = mx.log_softmax(lg)
    assert torch.equal(ls._t, lg._t - torch.logsumexp(lg._t.float(), -1, keepdim=True).to(torch.bfloat16))
    assert mx.where(mx.array([True, False]), 0, -mx.inf).dtype == f32
    o = mx.ones(2)
    o *= mx.array([True, False])
    assert o.dtype == f32 and o.tolist() == [1.0, 0.0]
