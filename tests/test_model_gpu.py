"""End-to-end parity on the GPU: the HIP model (through the C ABI) against the
CPU oracle on the same seeded weights and inputs.

Tolerance on bf16 logits: |diff| <= ATOL + RTOL*|ref| with RTOL = 2e-2,
ATOL = 1.25e-2*max|ref| is asserted on >= 99.9 % of entries and 4x that on all of
them; greedy tokens are asserted exact wherever the oracle's top-2 margin
exceeds 4*ATOL."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _mk(blind, seed=0, std_scale=4.0):
    from phi_3_vision_mlx_amd.api import load_synthetic
    import phi3v_oracle as orc
    model, proc = load_synthetic(blind_model=blind, tiny=True, seed=seed, std_scale=std_scale, device="cuda:0")
    oracle = orc.OraclePhi3V(model.cfg, {k: v.cpu() for k, v in model.w.items()}, cache_fp32=True)
    return model, proc, oracle


@pytest.fixture(scope="module")
def text():
    return _mk(True)


@pytest.fixture(scope="module")
def vis():
    return _mk(False)


def assert_logits(got, ref, what="", rel_atol=1.25e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    atol = rel_atol * ref.abs().max().item()
    err = (got - ref).abs()
    tol = atol + 2e-2 * ref.abs()
    frac_bad = (err > tol).float().mean().item()
    assert frac_bad <= 1e-3 and (err <= 4 * tol).all(), \
        f"{what}: {frac_bad:.5f} outside tol; max err {err.max():.4f}, |ref|max {ref.abs().max():.3f}"
    return atol


def assert_tokens_where_confident(got_logits, ref_logits, atol):
    ref = ref_logits.float().cpu()
    top2 = ref.topk(2, dim=-1).values
    confident = (top2[..., 0] - top2[..., 1]) > 4 * atol
    g, r = got_logits.float().cpu().argmax(-1), ref.argmax(-1)
    assert torch.equal(g[confident], r[confident])
    return confident.float().mean().item()


def rand_ids(n, seed):
    return np.random.default_rng(seed).integers(3, 32000, (1, n)).astype(np.int64)


def test_prefill_all_positions(text):
    model, _, oracle = text
    ids = rand_ids(150, 1)
    got, _ = model(input_ids=ids, max_tokens=4, full_logits=True)
    ref, _ = oracle(input_ids=ids, max_tokens=4)
    atol = assert_logits(got, ref, "prefill")
    assert assert_tokens_where_confident(got, ref, atol) > 0.1


def test_prefill_last_only_and_decode_teacher_forced(text):
    model, _, oracle = text
    ids = rand_ids(37, 2)
    n = 12
    got, cache = model(input_ids=ids, max_tokens=n)
    ref, oc = oracle(input_ids=ids, max_tokens=n)
    assert got.shape[1] == 1
    atol = assert_logits(got[:, -1], ref[:, -1], "prefill last")
    for step in range(n - 1):
        tok = torch.argmax(ref[:, -1].float(), dim=-1)[:, None]            # oracle's token feeds both
        got, cache = model(input_ids=tok, cache=cache)
        ref, oc = oracle(input_ids=tok, cache=oc)
        assert_logits(got[:, -1], ref[:, -1], f"decode step {step}")
        assert_tokens_where_confident(got[:, -1], ref[:, -1], atol)
    assert cache[0].offset == oc[0].offset == 37 + n - 1


def test_decode_equals_prefill_property(text):
    """Size-independent property: logits of token t from (prefill S, decode 1) == prefill S+1 last row."""
    model, _, _ = text
    ids = rand_ids(65, 3)
    a, cache = model(input_ids=ids[:, :64], max_tokens=2)
    b, _ = model(input_ids=ids[:, 64:], cache=cache)
    c, _ = model(input_ids=ids, max_tokens=1)
    assert_logits(b[:, -1], c[:, -1], "decode vs prefill")


def test_graph_replayed_greedy_step_equals_eager(text):
    """The hipGraph decode step (device-resident token/offset state) is bit-identical to the eager path."""
    model, proc, _ = text
    from phi_3_vision_mlx_amd import ops
    for inputs in (dict(input_ids=rand_ids(40, 7)), proc(["short", "a somewhat longer prompt here"])):
        n = 9
        lg, cache = model(**inputs, max_tokens=n)
        tok = ops.argmax(lg[:, -1, :].contiguous())[:, None]
        eager_tok, eager_lg, t = [], [], tok
        for _ in range(n - 1):
            lg, cache = model(input_ids=t, cache=cache, mask=inputs.get("mask"), pids=inputs.get("pids"))
            t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
            eager_tok.append(t.cpu()), eager_lg.append(lg[:, -1].cpu())
        lg, cache2 = model(**inputs, max_tokens=n)
        t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
        assert torch.equal(t.cpu(), tok.cpu())
        for i in range(n - 1):
            lg, t = model.greedy_step(t, cache2)
            assert torch.equal(lg[:, -1].cpu(), eager_lg[i]), f"step {i}"
            assert torch.equal(t.cpu(), eager_tok[i])
        assert cache2[0].offset == cache[0].offset
        hist = cache2[0].state.graphs["greedy"]["history"][:, :n - 1].cpu()
        assert torch.equal(hist, torch.cat(eager_tok, dim=1).to(hist.dtype))


def test_batched_left_pad(text):
    model, proc, oracle = text
    inputs = proc(["a", "hello world, this is a longer prompt", "mid size"])
    n = 5
    got, cache = model(**inputs, max_tokens=n)
    ref, oc = oracle(**inputs, max_tokens=n)
    assert_logits(got[:, -1], ref[:, -1], "batched prefill")
    for step in range(n - 1):
        tok = torch.argmax(ref[:, -1].float(), dim=-1)[:, None]
        got, cache = model(input_ids=tok, cache=cache, mask=inputs["mask"], pids=inputs["pids"])
        ref, oc = oracle(input_ids=tok, cache=oc, mask=inputs["mask"], pids=inputs["pids"])
        assert_logits(got[:, -1], ref[:, -1], f"batched decode {step}")
    # pad invariance: row 1 (the longest, no padding) equals its unbatched run
    solo, _ = model(**proc("hello world, this is a longer prompt"), max_tokens=n)
    again, _ = model(**inputs, max_tokens=n)
    assert_logits(again[1:2, -1], solo[:, -1], "pad invariance")


def test_vision_prefill_matches_oracle(vis):
    model, proc, oracle = vis
    from golden_inputs import make_image
    for (w, h, kind, seed) in [(336, 336, "noise", 0), (640, 480, "smooth", 1)]:
        inputs = proc("<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", [make_image(w, h, kind, seed)])
        got, cache = model(**inputs, max_tokens=3)
        ref, oc = oracle(**inputs, max_tokens=3)
        assert_logits(got[:, -1], ref[:, -1], f"vision prefill {w}x{h}")
        tok = torch.argmax(ref[:, -1].float(), dim=-1)[:, None]
        got, cache = model(input_ids=tok, cache=cache)
        ref, oc = oracle(input_ids=tok, cache=oc)
        assert_logits(got[:, -1], ref[:, -1], "vision decode")


def test_vision_tower_and_projector_stages(vis):
    """Stage-wise: ViT features (fp32) and the projected image embeddings vs the oracle."""
    model, proc, oracle = vis
    from golden_inputs import make_image
    inputs = proc("<|user|>\n<|image_1|>\nhi<|end|>\n<|assistant|>\n", [make_image(500, 1000, "noise", 2)])
    pv = torch.as_tensor(inputs["pixel_values"], dtype=F32)
    h, w = (np.asarray(inputs["image_sizes"])[0] // 336).tolist()
    live = h * w + 1
    feats = model.clip_forward(pv[0, :live].contiguous().cuda())[:, 1:].cpu()
    ref_feats = oracle.clip_model(pv[0, :live].cpu())
    err = (feats - ref_feats).abs().max().item()
    assert err <= 0.05 * ref_feats.abs().max().item() + 0.05, (err, ref_feats.abs().max().item())
    x = torch.zeros((inputs["input_ids"].shape[1], model.cfg.hidden_size), dtype=BF16)
    ref_x = oracle.image_embedding(x[None].clone(), inputs["pixel_values"], inputs["image_sizes"], inputs["positions"])[0]
    got_x = model.vision_embed(x.cuda(), inputs["pixel_values"], inputs["image_sizes"], inputs["positions"], x.shape[0]).cpu()
    rows = np.asarray(inputs["positions"])[:, 1]
    assert (got_x[rows].float() - ref_x[rows].float()).abs().max().item() <= 0.05 * ref_x.float().abs().max().item() + 0.02
    untouched = np.setdiff1d(np.arange(x.shape[0]), rows)
    assert got_x[untouched].abs().sum().item() == 0


def test_quantized_cache_close_to_bf16_cache():
    """quantize_cache=True (int8 KV): the prefill is exact (phi.py:531-533), decode logits stay within 3 % of max|logit|
    of the bf16-cache run, beams raise like the reference (phi.py:525)."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    mq, proc = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0", use_quantized_cache=True)
    mb, _ = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0")
    inputs = proc(["a short one", "a somewhat longer prompt for the second row of the batch"])
    n = 6
    lq, cq = mq(**inputs, max_tokens=n)
    lb, cb = mb(**inputs, max_tokens=n)
    assert cq[0].state.quantized and torch.equal(lq, lb)
    tok = model_tok(lb)
    for step in range(n - 1):
        lq, tq = mq.greedy_step(tok, cq)
        lb, tb = mb.greedy_step(tok, cb)
        err = (lq.float() - lb.float()).abs().max().item()
        assert err <= 3e-2 * lb.float().abs().max().item() + 1e-2, (step, err)
        tok = tb.clone()
    with pytest.raises(NotImplementedError):
        mq(input_ids=np.zeros((6, 3), dtype=np.int64), cache=cq, n_beam=3, advance_offset=0)


def test_fp8_weights_equal_bf16_model_on_dequantised_weights():
    """quantize_model=True: the fp8 model must equal a bf16 model that is handed the dequantised weights
    (isolates the kernels from the quantisation error itself), and stay close to the unquantised model."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.model import Phi3VModel
    from phi_3_vision_mlx_amd import ops
    mf, proc = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0", quantized_fp8=True)
    assert len(mf.w8) == 2 * 4 + 1 and "lm_head.weight" not in mf.w
    wd = dict(mf.w)
    for k, (w8, sc) in mf.w8.items():
        wd[k] = ops.dequant_fp8(w8, sc)
    mb = Phi3VModel(mf.cfg.__class__(**{**vars(mf.cfg), "quantized_fp8": False}), wd, device="cuda:0")
    m0, _ = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0")
    inputs = proc(["fp8 check", "a second and longer row for the batch"])
    lf, cf = mf(**inputs, max_tokens=5)
    lb, cb = mb(**inputs, max_tokens=5)
    l0, _ = m0(**inputs, max_tokens=5)
    assert_logits(lf[:, -1], lb[:, -1], "fp8 vs bf16-on-dequantised prefill")
    assert (lf.float() - l0.float()).abs().max().item() <= 0.2 * l0.float().abs().max().item()      # weight rounding, 3-bit mantissa
    tok = model_tok(lb)
    for step in range(4):
        lf, _ = mf.greedy_step(tok, cf)
        lb, tb = mb.greedy_step(tok, cb)
        assert_logits(lf[:, -1], lb[:, -1], f"fp8 vs bf16-on-dequantised decode {step}")
        tok = tb.clone()


def test_generate_choose_constrain_match_oracle_loops(text):
    """The public API on the HIP model vs the oracle's restatement of the same loops."""
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd import api
    model, proc, oracle = text
    prompts = ["<|user|>\nWhat is 2+2? A: 3 B: 4<|end|>\n<|assistant|>\n", "<|user|>\nName a colour.<|end|>\n<|assistant|>\n"]
    # choose
    got = api._choose_from(model, proc, prompts, "ABCDE", mute=True)
    options = proc([f" {c}" for c in "ABCDE"])["input_ids"][:, -1]
    ref_idx = orc.choose_from(oracle, proc(prompts), options)
    assert got == ["ABCDE"[i] for i in ref_idx]
    # greedy generate (B=2), EOS not expected with random weights: compare full token matrix
    n = 6
    inputs = proc(prompts)
    ref_tok, ref_lg = orc.greedy_generate(oracle, dict(inputs), n)
    logits, cache = model(**inputs, max_tokens=n)
    toks = [model_tok(logits)]
    for _ in range(n - 1):
        logits, cache = model(input_ids=toks[-1], cache=cache, mask=inputs["mask"], pids=inputs["pids"])
        toks.append(model_tok(logits))
    got_tok = torch.cat(toks, dim=1).cpu().long()
    top2 = ref_lg.float().topk(2, dim=-1).values
    if ((top2[..., 0] - top2[..., 1]) > 0.25).all():        # only assert exact when no near-tie occurred
        assert torch.equal(got_tok, ref_tok)
    # constrain (with and without beam): compare synthesised token rows and scores
    for use_beam in (False, True):
        idc = proc.tokenizer.encode(" The answer is", add_special_tokens=False)[1:]
        gs, gscore = api.constrain_tokens(model, dict(inputs), (4, " The answer is"), idc, use_beam=use_beam)
        rs, rscore = orc.constrain_one(oracle, dict(inputs), (4, " The answer is"), idc, use_beam=use_beam)
        assert gs.shape == rs.shape
        assert (gscore.float() - rscore.float()).abs().max().item() < 0.15, (gscore, rscore)
        if torch.equal(gs, rs) is False:
            # a near-tie may legitimately flip a greedy pick; the constraint tail must still be present
            tail = torch.tensor(idc)
            assert (gs[:, -len(idc):] == tail).all() or (gs[:, -len(idc) - 1:-1] == tail).all() or (gs == 32007).any()


def model_tok(logits):
    from phi_3_vision_mlx_amd import ops
    return ops.argmax(logits[:, -1, :].contiguous())[:, None]


def test_public_generate_runs_and_reports(text, capsys):
    from phi_3_vision_mlx_amd import api
    model, proc, _ = text
    out = api.generate("Say hi.", preload=(model, proc), max_tokens=5, verbose=False, stream=False)
    assert isinstance(out, list) and len(out) == 1 or isinstance(out, str)
    outs = api.generate(["Say hi.", "And bye, please."], preload=(model, proc), max_tokens=4, verbose=False)
    assert isinstance(outs, list) and len(outs) == 2
    tps = api.generate("Say hi.", preload=(model, proc), max_tokens=4, verbose=False, return_tps=True)
    assert len(tps) == 2 and tps[1] > 0
    with pytest.raises(ValueError):
        api.generate(["a", "b"], images=["x.png"], preload=(model, proc), apply_chat_template=False)
    txt = api.constrain("Pick one.", constraints=[(3, " The answer is"), "AB"], preload=(model, proc), verbose=False)
    assert isinstance(txt, str) and txt[-1] in "AB"


# ----------------------------------------------------------------------------- committed golden fixtures
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def _check_topk(logits, topv, topi, what, rel=2.5e-2):
    """HIP logits vs the fixture's top-16 (ids, values) of the oracle: values within `rel`*max|logit| at the
    oracle's top ids, and the argmax agrees whenever the oracle's top-2 margin exceeds twice that.
    rel: 2.5 % for the 2-layer tiny model, 6 % for the 32-layer model -- tools/precision_study.py shows that ONE
    bf16 rounding anywhere in the attention block (the reference keeps q, k, P, o in fp32) already moves the
    final bf16 logits by 2-3 % of their maximum after 24 layers: the bf16 residual stream amplifies it."""
    lg = logits.float().cpu()
    ref_v, ref_i = torch.as_tensor(topv), torch.as_tensor(topi).long()
    got_v = torch.gather(lg, -1, ref_i)
    scale = ref_v.abs().max().item()
    err = (got_v - ref_v).abs().max().item()
    assert err <= rel * scale + 1e-2, f"{what}: top-k logit error {err:.4f} (scale {scale:.2f})"
    margin = ref_v[..., 0] - ref_v[..., 1]
    clear = margin > 2 * rel * scale
    assert torch.equal(lg.argmax(-1)[clear], ref_i[..., 0][clear]), what
    return clear


def test_tiny_fixture_text_and_batch(text):
    model, proc, _ = text
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    from phi_3_vision_mlx_amd import ops
    for key, inputs in (("text", {"input_ids": g["text_ids"]}),
                        ("batch", proc(["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n",
                                        "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]))):
        ref_tok = torch.as_tensor(g[f"{key}_tokens"]).long()
        n = ref_tok.shape[1]
        logits, cache = model(**inputs, max_tokens=n)
        for step in range(n):                                   # teacher-forced with the fixture's tokens
            _check_topk(logits[:, -1], g[f"{key}_topv"][:, step], g[f"{key}_topi"][:, step], f"{key} step {step}")
            if step + 1 < n:
                logits, cache = model(input_ids=ref_tok[:, step:step + 1], cache=cache, mask=inputs.get("mask"), pids=inputs.get("pids"))


def test_tiny_fixture_vision(vis):
    model, proc, _ = vis
    from golden_inputs import make_image
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    inputs = proc("<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", [make_image(336, 336, "noise", 0)])
    assert np.asarray(inputs["input_ids"]).shape[1] == int(g["vis_n_ids"][0])
    ref_tok = torch.as_tensor(g["vis_tokens"]).long()
    logits, cache = model(**inputs, max_tokens=4)
    for step in range(4):
        _check_topk(logits[:, -1], g["vis_topv"][:, step], g["vis_topi"][:, step], f"vision step {step}")
        if step + 1 < 4:
            logits, cache = model(input_ids=ref_tok[:, step:step + 1], cache=cache)


def test_synthetic_weights_identical_on_gpu_and_cpu():
    from phi_3_vision_mlx_amd.weights import synth_values
    a, b = synth_values(300001, 12345, 0.02, device="cpu"), synth_values(300001, 12345, 0.02, device="cuda:0")
    assert torch.equal(a, b.cpu())


@pytest.fixture(scope="module")
def full_text():
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, proc = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0")
    yield model, proc
    del model
    torch.cuda.empty_cache()


def test_c1_fixture_full_size(full_text):
    """BASELINE config 1 (Phi-3-mini-128K, 128-token prompt, greedy) at FULL size vs the oracle fixture
    generated on CPU (tests/golden/gen_golden_oracle.py c1): same hash-seeded weights on both sides."""
    model, _ = full_text
    g = np.load(GOLDEN + "/c1_oracle.npz")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    n = ref_tok.shape[1]
    logits, cache = model(input_ids=g["ids"], max_tokens=n)
    n_clear = 0
    for step in range(n):
        n_clear += int(_check_topk(logits[:, -1], g["topv"][:, step], g["topi"][:, step], f"C1 step {step}", rel=6e-2).sum())
        if step + 1 < n:
            logits, tok = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    # rank agreement where it is meaningful: the oracle's top-1 must be inside the HIP top-16 at every step
    last = logits[:, -1].float().cpu().topk(16).indices[0].tolist()
    assert int(g["topi"][0, n - 1, 0]) in last


def test_c2_fixture_full_size_vision():
    """BASELINE config 2 = bench.py's rank-0 request (one seeded 336x336 image -> 17 CLIP crops -> 2509 image tokens,
    2531-token prompt) at FULL size vs the oracle fixture (tests/golden/gen_golden_oracle.py c2): CLIP tower,
    projector, HD merge, 32 decoder layers and 3 graph-replayed decode steps, same hash-seeded weights on both sides."""
    import sys
    sys.path.insert(0, GOLDEN)
    from gen_golden_oracle import c2_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, proc = load_synthetic(blind_model=False, tiny=False, seed=0, device="cuda:0")
    g = np.load(GOLDEN + "/c2_oracle.npz")
    inp = c2_request(proc.img_processor)
    assert inp["input_ids"].shape[1] == int(g["n_ids"][0]) == 2531
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    n = ref_tok.shape[1]
    logits, cache = model(**inp, max_tokens=n)
    for step in range(n):
        _check_topk(logits[:, -1], g["topv"][:, step], g["topi"][:, step], f"C2 step {step}", rel=6e-2)
        top16 = logits[:, -1].float().cpu().topk(16).indices[0].tolist()
        assert int(g["topi"][0, step, 0]) in top16, f"C2 step {step}: oracle top-1 not in the HIP top-16"
        if step + 1 < n:
            logits, tok = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    del model, cache
    torch.cuda.empty_cache()


def test_config3_long_context_32k(full_text):
    """BASELINE config 3: 32k-token prompt (Su/LongRoPE long factors, chosen once from S+max_tokens > 4096, Q2).
    No CPU oracle can run this size; parity is checked through size-independent properties:
    (a) the RoPE table the model built equals the long-factor formula, (b) prefill(S)+decode(1) == prefill(S+1),
    (c) the graph-replayed step equals the eager step bit for bit at this length."""
    from phi_3_vision_mlx_amd.config import LONG_FACTOR, rope_scaling_factor
    model, _ = full_text
    S = 32768
    ids = np.random.default_rng(4).integers(3, 32000, (1, S + 1)).astype(np.int64)
    a, cache = model(input_ids=ids[:, :S], max_tokens=4)
    st = cache[0].state
    for pos in (0, 4097, S - 1):
        e = pos / (np.asarray(LONG_FACTOR) * 10000.0 ** (np.arange(0, 96, 2) / 96))
        ref = np.cos(e) * rope_scaling_factor(model.cfg)
        assert np.allclose(st.cos[0, pos].cpu().numpy(), ref, atol=2e-5 + pos * 3e-7), pos
    b, _ = model(input_ids=ids[:, S:], cache=cache)
    st.offset = S                                               # rewind; replay the same step through the graph
    g, _ = model.greedy_step(torch.as_tensor(ids[:, S:]).to("cuda:0", torch.int32), cache)
    assert torch.equal(g[:, -1], b[:, -1])
    del cache
    torch.cuda.empty_cache()
    c, _ = model(input_ids=ids, max_tokens=1)
    assert_logits(b[:, -1], c[:, -1], "32k decode vs prefill", rel_atol=6e-2)


def test_full_size_decode_equals_prefill_property(full_text):
    """Size-independent property at full size: (prefill S, decode 1) == (prefill S+1) on the last row."""
    model, _ = full_text
    ids = rand_ids(300, 9)
    a, cache = model(input_ids=ids[:, :299], max_tokens=2)
    b, _ = model(input_ids=ids[:, 299:], cache=cache)
    c, _ = model(input_ids=ids, max_tokens=1)
    assert_logits(b[:, -1], c[:, -1], "full-size decode vs prefill", rel_atol=6e-2)   # 32 layers: see _check_topk


def _synth_adapter(cfg, targets, layers, rank, seed=5):
    """A 'trained' adapter: lora_a ~ U(-1/sqrt(in), 1/sqrt(in)) as LoRALinear.__init__ (phi.py:121-126), lora_b non-zero."""
    gen = torch.Generator().manual_seed(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size
    qkv = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * (H // cfg.num_attention_heads)
    dims = {"self_attn.qkv_proj": (H, qkv), "self_attn.o_proj": (H, H), "mlp.gate_up_proj": (H, 2 * I), "mlp.down_proj": (I, H)}
    idx = list(range(cfg.num_hidden_layers))[-layers:] if isinstance(layers, int) else layers
    tensors = {}
    for i in idx:
        for t in targets:
            k_in, k_out = dims[t]
            tensors[f"model.layers.{i}.{t}.lora_a"] = (torch.rand((k_in, rank), generator=gen) * 2 - 1) * k_in ** -0.5
            tensors[f"model.layers.{i}.{t}.lora_b"] = torch.randn((rank, k_out), generator=gen) * 0.04
    lora_cfg = {"model_path": "models/x", "adapter_path": "adapters/x", "lora_layers": layers, "lora_targets": targets,
                "lora_parameters": {"rank": rank, "alpha": 2 * rank, "dropout": 0.0, "scale": 1.5}}
    return lora_cfg, tensors


@pytest.mark.parametrize("targets,layers,rank", [(["self_attn.qkv_proj"], 1, 1),
                                                 (["self_attn.qkv_proj", "self_attn.o_proj", "mlp.gate_up_proj", "mlp.down_proj"], [0, 1], 8)])
def test_lora_adapter_matches_oracle(tmp_path, targets, layers, rank):
    """use_adapter=True: adapter files in the reference's format -> HIP model vs the oracle's LoRALinear restatement,
    prefill + eager cached call + graph-replayed decode steps; and the adapter must actually change the logits."""
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.weights import load_adapter, resolve_adapter, save_adapter
    base, _ = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0")
    lora_cfg, tensors = _synth_adapter(base.cfg, targets, layers, rank)
    save_adapter(str(tmp_path / "ad"), lora_cfg, tensors)
    model, proc = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0", adapter_path=str(tmp_path / "ad"))
    assert len(model.adapters) == len(tensors) // 2
    oracle = orc.OraclePhi3V(model.cfg, {k: v.cpu() for k, v in model.w.items()}, cache_fp32=True,
                             adapters=resolve_adapter(model.cfg, *load_adapter(str(tmp_path / "ad"))))
    ids = np.random.default_rng(21).integers(3, 32000, (2, 33)).astype(np.int64)
    n = 5
    ref_tok, ref_lg = orc.greedy_generate(oracle, {"input_ids": ids}, n, stop_on_eos=False)
    logits, cache = model(input_ids=ids, max_tokens=n)
    plain, _ = base(input_ids=ids, max_tokens=n)
    assert (logits.float() - plain.float()).abs().max().item() > 0.05       # the adapter is live
    for step in range(n):
        # 2 % instead of 1.25 %: an adapted o_proj runs on the fp32 attention output in the reference (one rounding at the
        # end); here both the attention output and the frozen projection are already bf16 when the rank-r term is added
        assert_logits(logits[:, -1], ref_lg[:, step], f"lora step {step}", rel_atol=2e-2)
        if step + 1 < n:
            tok = ref_tok[:, step:step + 1].to("cuda:0", torch.int32)
            if step == 0:
                logits, cache = model(input_ids=tok, cache=cache)            # eager cached call
            else:
                logits, _ = model.greedy_step(tok, cache)                   # graph replay


def test_int4_weights_full_size_decode_matches_dequantised_model():
    """quantize_model (4-bit group-64, the reference's nn.quantize format): a full-size model running p3v_gemv_q4 in
    decode and dequantise-to-scratch in prefill vs THE SAME model with its 4-bit weights expanded to bf16 up front
    (scale*q+bias rounded once): logits within the 32-layer tolerance, greedy tokens equal where the margin is clear."""
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.model import Phi3VModel
    mq, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", quantized_int4=True)
    assert mq.w4 and "lm_head.weight" in mq.w4 and "lm_head.weight" not in mq.w
    wd = dict(mq.w)
    for k, (w4, sb) in mq.w4.items():
        wd[k] = ops.dequant_q4(w4, sb)
    cfg = type(mq.cfg)(**{k: v for k, v in vars(mq.cfg).items() if k != "quantized_int4"})
    md = Phi3VModel(cfg, wd, device="cuda:0")
    ids = np.random.default_rng(5).integers(3, 32000, (1, 200)).astype(np.int64)
    lq, cq = mq(input_ids=ids, max_tokens=6)
    ld, cd = md(input_ids=ids, max_tokens=6)
    assert_logits(lq[:, -1], ld[:, -1], "int4 prefill")          # same bf16 GEMMs on the same dequantised values; only the
                                                                # last-row lm_head differs (4-bit GEMV vs bf16 GEMV)
    for step in range(5):
        tok = ops.argmax(ld[:, -1].contiguous())[:, None]
        lq, _ = mq.greedy_step(tok, cq)
        ld, _ = md.greedy_step(tok, cd)
        assert_logits(lq[:, -1], ld[:, -1], f"int4 decode step {step}", rel_atol=6e-2)
    del mq, md
    torch.cuda.empty_cache()
