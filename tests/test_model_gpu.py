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


def assert_logits(got, ref, what="", rel_atol=1.5e-2):
    """(1.25e-2 until round 4: RMSNorm now rounds twice on both sides, as mx.fast.rms_norm does -- one more bf16 rounding per
    norm at which two correct implementations can part by an ulp; the tiny model's worst step went from 0.9 to 1.2 x the
    old bound.)"""
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


def test_generate_loop_survives_a_timed_out_fused_launch(monkeypatch, capsys):
    """The fused attention + o_proj launch needs the GPU to itself; when another process starves it the step's row is poisoned and
    the token comes back negative.  `_generate`'s loop then re-plans the step without that launch (as a server-owned model does),
    rewinds to the last good token and goes on: same tokens, same cache offset (the failure is injected with P3V_DEBUG_FAIL_STEP)."""
    from phi_3_vision_mlx_amd import api
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, processor = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", num_hidden_layers=2)
    ids = torch.randint(3, 32000, (1, 1700), dtype=torch.int64, generator=torch.Generator().manual_seed(6))

    class Keep:
        def __init__(self): self.rows = []
        def __call__(self, rows): self.rows.append(list(rows))

    def run(fail):
        if fail is None:
            monkeypatch.delenv("P3V_DEBUG_FAIL_STEP", raising=False)
        else:
            monkeypatch.setenv("P3V_DEBUG_FAIL_STEP", str(fail))
        logits, cache = model(input_ids=ids, max_tokens=40)
        token = ops_mod.argmax(logits[:, -1].contiguous())[:, None]
        keep = Keep()
        out = api.greedy_loop(model, token, cache, 12, keep, lambda rows: False)
        return keep.rows, out.reshape(-1).tolist(), cache[0].state.offset, cache[0].state.graphs["greedy"]["bufs"].get("fuse_o", False)

    from phi_3_vision_mlx_amd import ops as ops_mod
    model.serving = False
    good, last, off, fused = run(None)
    assert fused and len(good) == 12 and off == 1700 + 12
    for fail in (0, 5, 11):
        model.serving = False
        rows, last2, off2, fused2 = run(fail)
        assert rows == good and last2 == last and off2 == off and not fused2 and model.serving
        assert "continuing with separate launches" in capsys.readouterr().err
    model.serving = True                                          # already degraded: the same failure is raised
    with pytest.raises(RuntimeError, match="device step failed"):
        run(3)
    del model
    torch.cuda.empty_cache()


def test_short_prompt_resid_norm_fusion_changes_nothing(monkeypatch):
    """A 128-token prompt runs o_proj / down_proj as K slices whose reduction launch also writes the next RMSNorm
    (model._proj_resid_norm): logits and cache equal the unfused launches bit for bit."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", num_hidden_layers=3)
    ids = torch.randint(3, 32000, (1, 128), dtype=torch.int64, generator=torch.Generator().manual_seed(5))
    monkeypatch.setenv("P3V_PREFILL_GRAPH", "0")
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("P3V_RESID_NORM_FUSE", flag)
        logits, cache = model(input_ids=ids, max_tokens=4)
        st = cache[0].state
        outs.append((logits.clone(), st.k[:, :, :, :128].clone(), st.v[:, :, :, :, :128].clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    del model
    torch.cuda.empty_cache()


def test_captured_short_prompt_prefill_equals_eager(text, monkeypatch):
    """Round 5: a text prompt whose (length, max_tokens) was seen before is prefilled by ONE hipGraph over buffers the entry owns
    (model._prefill_captured).  Same kernels in the same order: logits, cache contents and the greedy continuation are bit-identical to
    the eager path; the entry's cache is leased -- while the caller keeps it, the next prompt of that geometry gets a state of its own
    (eager), and once it is dropped the entry is reused."""
    from phi_3_vision_mlx_amd import ops
    model, _, _ = text
    model._prefill_graphs.clear(), model._prefill_seen.clear()
    ids_a, ids_b = rand_ids(57, 41), rand_ids(57, 42)
    monkeypatch.setenv("P3V_PREFILL_GRAPH", "0")
    ref = {}
    for name, ids in (("a", ids_a), ("b", ids_b)):
        lg, cache = model(input_ids=ids, max_tokens=6)
        st = cache[0].state
        toks, t = [], ops.argmax(lg[:, -1, :].contiguous())[:, None]
        for _ in range(4):
            _, t = model.greedy_step(t, cache)
            toks.append(t.clone())
        ref[name] = (lg.clone(), st.k[:, :, :, :57].clone(), st.v[..., :57].clone(), torch.cat(toks, 1).clone())
        del cache, st
    monkeypatch.setenv("P3V_PREFILL_GRAPH", "1")
    l1, c1 = model(input_ids=ids_a, max_tokens=6)                # first sighting of (57, 6): eager
    assert not model._prefill_graphs
    del c1
    l2, c2 = model(input_ids=ids_a, max_tokens=6)                # second: captured and replayed
    assert len(model._prefill_graphs) == 1 and next(iter(model._prefill_graphs))[:2] == (57, 6)
    entry_state = next(iter(model._prefill_graphs.values()))["st"]
    assert c2[0].state is not entry_state and c2[0].state.k.data_ptr() == entry_state.k.data_ptr()    # its own state object over the entry's buffers
    assert torch.equal(l2, ref["a"][0]) and torch.equal(entry_state.k[:, :, :, :57], ref["a"][1]) and torch.equal(entry_state.v[..., :57], ref["a"][2])
    l3, c3 = model(input_ids=ids_b, max_tokens=6)                # c2 still alive: the entry is leased -> a state of its own
    assert c3[0].state.k.data_ptr() != entry_state.k.data_ptr() and torch.equal(l3, ref["b"][0])
    t = ops.argmax(l2[:, -1, :].contiguous())[:, None]           # the leased cache is intact and decodes as the eager one did
    toks = []
    for _ in range(4):
        _, t = model.greedy_step(t, c2)
        toks.append(t.clone())
    assert torch.equal(torch.cat(toks, 1), ref["a"][3])
    del c2, c3
    l4, c4 = model(input_ids=ids_b, max_tokens=6)                # lease released: the entry again, other prompt, decode graph reused
    assert c4[0].state.k.data_ptr() == entry_state.k.data_ptr() and torch.equal(l4, ref["b"][0])
    t = ops.argmax(l4[:, -1, :].contiguous())[:, None]
    toks = []
    for _ in range(4):
        _, t = model.greedy_step(t, c4)
        toks.append(t.clone())
    assert torch.equal(torch.cat(toks, 1), ref["b"][3])
    # the lease is the STATE, not the list: a caller that keeps `cache[0].state` and drops the list still owns the buffers
    st4 = c4[0].state
    del c4
    l6, c6 = model(input_ids=ids_a, max_tokens=6)
    assert c6[0].state.k.data_ptr() != entry_state.k.data_ptr() and torch.equal(l6, ref["a"][0])
    assert torch.equal(st4.k[:, :, :, :57], ref["b"][1]) and torch.equal(st4.v[..., :57], ref["b"][2]) and st4.offset == 57 + 4
    del st4, c6
    # an entry whose caller died in an unrecovered failed step is dropped, and the geometry captures afresh
    l7, c7 = model(input_ids=ids_a, max_tokens=6)
    assert c7[0].state.k.data_ptr() == entry_state.k.data_ptr()
    c7[0].state.mark_dirty()
    del c7
    l8, c8 = model(input_ids=ids_a, max_tokens=6)                # dirty: dropped, eager
    assert not model._prefill_graphs and torch.equal(l8, ref["a"][0])
    del c8
    l9, c9 = model(input_ids=ids_b, max_tokens=6)                # captured again on fresh buffers
    assert len(model._prefill_graphs) == 1 and torch.equal(l9, ref["b"][0])
    del c9
    big = model.PREFILL_GRAPH_MAX_CACHE_BYTES
    try:                                                          # the entry's cache is capped
        type(model).PREFILL_GRAPH_MAX_CACHE_BYTES = 1
        model._prefill_graphs.clear(), model._prefill_seen.clear()
        for _ in range(3):
            lx, cx = model(input_ids=ids_a, max_tokens=6)
            del cx
        assert not model._prefill_graphs and not model._prefill_seen and torch.equal(lx, ref["a"][0])
    finally:
        type(model).PREFILL_GRAPH_MAX_CACHE_BYTES = big
    l5, _ = model(input_ids=rand_ids(58, 43), max_tokens=6)      # another length: eager (first sighting), nothing captured for it
    assert all(k[:2] != (58, 6) for k in model._prefill_graphs) and torch.isfinite(l5.float()).all()
    model._prefill_graphs.clear(), model._prefill_seen.clear()


def test_captured_short_prompt_prefill_with_4bit_weights_equals_eager(monkeypatch):
    """Round 6: the captured short-prompt prefill also takes MLX 4-bit checkpoints (each projection = a dequantise launch into the model's
    own scratch + the bf16 GEMM: the same launches, captured): logits and greedy continuation bit-identical to the eager path."""
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", num_hidden_layers=2, quantized_int4=True)
    ids = rand_ids(57, 43)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("P3V_PREFILL_GRAPH", mode)
        for rep in range(3):                                     # (mode 1: eager, capture, replay)
            lg, cache = model(input_ids=ids, max_tokens=6)
            t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
            toks = []
            for _ in range(3):
                _, t = model.greedy_step(t, cache)
                toks.append(t.clone())
            got = (lg.clone(), torch.cat(toks, 1).clone())
            if mode == "0" and rep == 0:
                out["ref"] = got
            assert torch.equal(got[0], out["ref"][0]) and torch.equal(got[1], out["ref"][1]), (mode, rep)
            del cache
    assert len(model._prefill_graphs) == 1


def test_graph_decode_with_more_than_16_rows(text):
    """B = 20 rows: the decode projections are split-K GEMMs (M > 16), whose workspace the captured graph must own (ADVICE r03:
    no allocation under capture, no dangling pointer after a regrow).  Graph replays == eager steps, bit for bit, also after
    another shape has regrown the shared per-stream workspace in between."""
    model, proc, _ = text
    from phi_3_vision_mlx_amd import ops
    ids = np.random.default_rng(3).integers(3, 32000, (20, 24)).astype(np.int64)
    n = 6
    lg, cache = model(input_ids=ids, max_tokens=n)
    tok = ops.argmax(lg[:, -1, :].contiguous())[:, None]
    eager, t = [], tok
    for _ in range(n - 1):
        lg, cache = model(input_ids=t, cache=cache)
        t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
        eager.append((t.cpu(), lg[:, -1].cpu()))
    lg, cache2 = model(input_ids=ids, max_tokens=n)
    t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
    for i in range(n - 1):
        if i == 2:                                               # a bigger split-K problem on the same stream in between
            big = torch.randn((700, model.cfg.hidden_size), device="cuda:0").to(BF16)
            ops.gemm(big, model.w["model.layers.0.mlp.gate_up_proj.weight"], ops.EPI_SILU_MUL)
        lg, t = model.greedy_step(t, cache2)
        assert torch.equal(t.cpu(), eager[i][0]) and torch.equal(lg[:, -1].cpu(), eager[i][1]), f"step {i}"
    g = cache2[0].state.graphs["greedy"]
    assert "gemm_ws" in g


def test_split_merge_launch_is_a_tested_fallback_of_the_in_launch_merge(text, monkeypatch):
    """the in-launch split-KV merge relies on workgroups being dispatched in linear order (true on this runtime,
    not a HIP guarantee) -- `P3V_ATTN_FUSED_MERGE=0` takes the separate merge launch instead.  Both plans must produce the
    same greedy tokens and logits that agree to bf16 rounding (same partials, different summation order of the splits),
    through the captured decode graph, at a context with several splits."""
    model, proc, _ = text
    from phi_3_vision_mlx_amd import ops
    inputs, n = dict(input_ids=rand_ids(700, 17)), 6
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("P3V_ATTN_FUSED_MERGE", mode)
        lg, cache = model(**inputs, max_tokens=n)
        t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
        toks, lgs = [t.cpu()], []
        for _ in range(n - 1):
            lg, t = model.greedy_step(t, cache)
            toks.append(t.cpu().clone()), lgs.append(lg[:, -1].float().cpu().clone())
        assert cache[0].state.graphs["greedy"]["bufs"]["attn_merge"] == (mode == "1")
        runs[mode] = (torch.cat(toks, 1), torch.stack(lgs))
    assert torch.equal(runs["1"][0], runs["0"][0])
    assert_logits(runs["1"][1].flatten(0, 1), runs["0"][1].flatten(0, 1), "merge in launch vs merge launch", rel_atol=1e-2)


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
    lq, cq = mq(**inputs, max_tokens=n + 24)
    lb, cb = mb(**inputs, max_tokens=n + 24)
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
    # a cached call with MORE than 16 new tokens (constrain() with a long constraint text, phi_3_vision_mlx.py:545): the
    # int8 layer is dequantised and attended through the prefill kernel, the new rows are quantised behind it
    long_ids = np.random.default_rng(3).integers(3, 32000, (2, 21)).astype(np.int64)
    off = cq[0].offset
    lq, _ = mq(input_ids=long_ids, cache=cq, advance_offset=0)
    lb, _ = mb(input_ids=long_ids, cache=cb, advance_offset=0)
    assert cq[0].offset == off and lq.shape == lb.shape == (2, 21, mq.cfg.vocab_size)
    assert (lq.float() - lb.float()).abs().max().item() <= 3e-2 * lb.float().abs().max().item() + 1e-2


def test_mlx4_prompt_cache_is_the_references_semantics():
    """load(..., quantize_cache=True, cache_format="mlx4"): the reference's own quantised cache (phi.py:528-540).  (a) the prefill
    attends on the EXACT keys: its logits equal the bf16-cache model's bit for bit; (b) from the second call on the prompt's K / V are
    mx.dequantize(mx.quantize(., group 32, 4 bits)) and later tokens stay unquantised: V rows equal the host round trip of the bf16
    model's rows (weights.mlx_quantize / mlx_dequantize), K rows are the round trip of the keys' exact fp32 values (pinned by
    test_kv_quantize_mlx4_keys_from_exact_fp32_rotation and the reference fixture `q4cache`) -- within one 4-bit step of the bf16
    rows' round trip -- and the decode steps are bit-identical to a bf16-cache model holding the same rows; (c) beam reads of the
    quantised cache raise as in the reference (phi.py:525)."""
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    plain, _ = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0")
    q4, _ = load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0", use_quantized_cache=True, cache_format="mlx4")
    ids = rand_ids(70, 31)
    la, ca = plain(input_ids=ids, max_tokens=8)
    lb, cb = q4(input_ids=ids, max_tokens=8)
    assert torch.equal(la, lb), "the prefill must attend on the exact keys"
    sa, sb = ca[0].state, cb[0].state
    assert sb.mlx4 and not sb.quantized and sb.mlx4_tokens == 70
    S, hd = 70, plain.hd
    rows = sa.v.transpose(3, 4)[:, :, :, :S].float().cpu()                           # [nl, B, nkv, S, hd]
    nl, B, nkv = rows.shape[:3]
    deq = mlx_dequantize(*mlx_quantize(rows.to(BF16).reshape(nl * B * nkv, S * hd), 32, 4), 32, 4, dtype=BF16).reshape(nl, B, nkv, S, hd).cuda()
    assert torch.equal(sb.v[..., :S], deq.transpose(3, 4)), "V rows are not mx.dequantize(mx.quantize(V)) of the bf16 values"
    krt = sa.k[:, :, :, :S].float().cpu()
    kdq = mlx_dequantize(*mlx_quantize(krt.reshape(nl * B * nkv, S * hd), 32, 4), 32, 4).reshape(nl, B, nkv, S, hd)
    step = (krt.reshape(nl, B, nkv, S, 3, 32).amax(-1) - krt.reshape(nl, B, nkv, S, 3, 32).amin(-1)) / 15      # one 4-bit code of each group
    assert ((sb.k[:, :, :, :S].float().cpu() - kdq).abs().reshape(nl, B, nkv, S, 3, 32) <= 2.1 * step[..., None] + 1e-3).all()
    sa.v[..., :S] = sb.v[..., :S]
    sa.k[:, :, :, :S] = sb.k[:, :, :, :S]
    tok = ops.argmax(la[:, -1, :].contiguous())[:, None]
    ta, tb = tok, tok
    for step_i in range(6):
        la, ta = plain.greedy_step(ta, ca)
        lb, tb = q4.greedy_step(tb, cb)
        assert torch.equal(la, lb) and torch.equal(ta, tb), f"decode step {step_i} differs"
    with pytest.raises(NotImplementedError):
        q4(input_ids=np.asarray([[5, 6]]), cache=cb, n_beam=2, advance_offset=0)


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


def walk_decisions(got, ref, clear):
    """Two decision traces of the same loop (api.constrain_tokens / oracle.constrain_one): every decision must have the
    same outcome until the first one whose ORACLE margin is below `clear` (a near-tie may legitimately flip and the runs
    part ways there).  Returns (number of agreeing decisions, diverged?)."""
    n = 0
    for (gk, gout), (rk, margin, rout) in zip(got, ref):
        assert gk == rk, (n, gk, rk)
        if gout != rout:
            assert margin <= clear, f"decision {n} ({rk}) differs at a clear margin {margin:.3f}: {gout} vs {rout}"
            return n, True
        n += 1
    assert len(got) == len(ref)
    return n, False


def test_generate_choose_constrain_match_oracle_loops(text):
    """The public API's loops on the HIP model vs the oracle's restatement of the same loops (phi_3_vision_mlx.py:376-409,
    466-487, 500-619), under a decisive lm_head: tokens / picks are compared exactly, decision by decision; a difference
    is accepted only at a decision whose oracle margin is below 4 x the logit tolerance, and must not happen early."""
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd import api
    from phi_3_vision_mlx_amd.weights import peaked_lm_head
    model, proc, oracle = text
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    clear = 4 * float(g["rel_tol"][0])                         # decision margins are fractions of max|logit| of their forward
    base_dev, base_cpu = model.w["lm_head.weight"], oracle.w["lm_head.weight"]

    def with_head(hs):
        model.w["lm_head.weight"] = peaked_lm_head(base_dev, float(g["spread"][0]), int(hs))
        oracle.w["lm_head.weight"] = peaked_lm_head(base_cpu, float(g["spread"][0]), int(hs))
        oracle._f32.pop("lm_head.weight", None)
    prompts = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
    try:
        # choose: the fixture's seed makes the option margins clear -> exact
        with_head(g["choose_head_seed"][0])
        got = api._choose_from(model, proc, prompts, "ABCDE", mute=True)
        assert got == ["ABCDE"[i] for i in g["choose_idx"]]
        # greedy generate through the public loop (B = 2, graph-replayed steps): exact token matrix
        with_head(g["batch_head_seed"][0])
        n = g["batch_tokens"].shape[1]
        inputs = proc(prompts)
        logits, cache = model(**inputs, max_tokens=n)
        toks = [model_tok(logits)]
        for _ in range(n - 1):
            logits, cache = model(input_ids=toks[-1], cache=cache, mask=inputs["mask"], pids=inputs["pids"])
            toks.append(model_tok(logits))
        assert torch.equal(torch.cat(toks, dim=1).cpu().long(), torch.as_tensor(g["batch_tokens"]).long())
        # constrain (with and without beam): same decisions as a live oracle run
        idc = proc.tokenizer.encode(" The answer is", add_special_tokens=False)[1:]
        checked = 0
        for use_beam in (False, True):
            for hs in (0, 1, 2):
                with_head(hs)
                cin = dict(inputs) if not use_beam else proc(prompts[:1] * 2)
                gt, rt = [], []
                gs, gscore = api.constrain_tokens(model, dict(cin), (3, " The answer is"), idc, use_beam=use_beam, trace=gt)
                rs, rscore = orc.constrain_one(oracle, dict(cin), (3, " The answer is"), idc, use_beam=use_beam, trace=rt)
                n_same, diverged = walk_decisions(gt, rt, clear)
                checked += n_same
                if not diverged:
                    assert torch.equal(gs, rs), (gs, rs)
                    scale = max(abs(float(rscore.float().min())), 1.0)
                    assert (gscore.float() - rscore.float()).abs().max().item() <= 0.03 * scale, (gscore, rscore)
        assert checked >= 40, checked                           # 6 runs x (7 | 19) decisions: most of them are compared
    finally:
        model.w["lm_head.weight"] = base_dev
        oracle.w["lm_head.weight"] = base_cpu
        oracle._f32.pop("lm_head.weight", None)


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


def test_benchmark_like_the_reference(tmp_path, monkeypatch, capsys):
    """benchmark(): three tasks x four variants (vanilla / quantize_model / quantize_cache / LoRA), JSON layout and table
    as the reference's (phi_3_vision_mlx.py:1178-1277, 427-443); tiny synthetic weights, 6 new tokens."""
    import json
    from phi_3_vision_mlx_amd import api
    monkeypatch.chdir(tmp_path)
    res = api.benchmark(json_path=str(tmp_path / "b.json"), synthetic="tiny", max_tokens=6)
    assert list(res) == ["vanilla", "q_model", "q_cache", "lora"]
    for rows in res.values():
        assert [r[0] for r in rows] == [0, 1, 2] and all(r[1] > 0 and r[2] > 0 for r in rows)
    assert json.load(open(tmp_path / "b.json")) == res
    out = capsys.readouterr().out
    assert "| Batched Generation" in out and "Quantized Cache" in out and out.count(" tps") == 12


# ----------------------------------------------------------------------------- committed golden fixtures
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def _from_bits(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(torch.bfloat16).float()


def head_row_norms(model):
    """fp32 L2 norms of the lm_head rows the model really multiplies by (bf16, or e4m3 x scale for quantize_model)."""
    if "lm_head.weight" in model.w:
        return model.w["lm_head.weight"].float().norm(dim=-1).cpu().clamp_min(1e-30)
    if "lm_head.weight" in getattr(model, "w4", {}):                 # 4-bit head: the values scale * q + bias the GEMV multiplies by
        from phi_3_vision_mlx_amd import ops
        return ops.dequant_q4(*model.w4["lm_head.weight"]).float().norm(dim=-1).cpu().clamp_min(1e-30)
    w8, sc = model.w8["lm_head.weight"]
    return (w8.view(torch.float8_e4m3fn).float().norm(dim=-1) * sc).cpu().clamp_min(1e-30)


def logits_vs_fixture(logits, g, step, norms, what, prefix=""):
    """HIP last-position logits vs the fixture's FULL oracle logits of `step` under the fixture's tolerance model
    (tests/golden/gen_golden_oracle.py): with z = logit / lm_head row norm, EVERY vocabulary entry within
    rel_tol x max|z| x its row norm (+ one bf16 ulp of the entry).  Returns (got, per-row clear flags, worst error in units
    of the tolerance): a step is clear when the oracle's top-2 margin exceeds the sum of the two entries' tolerances --
    then an implementation inside the tolerance cannot pick another token."""
    rel_tol = torch.as_tensor(g["rel_tol"]).float()
    ref = _from_bits(g[prefix + "logits_bf16"][:, step])
    got = logits.float().cpu().reshape(ref.shape)
    E = (rel_tol * (ref / norms).abs().amax(-1))[:, None]
    err = (got - ref).abs()
    worst = ((err - 2.0 ** -7 * ref.abs()).clamp_min(0) / (E * norms)).max().item()
    assert worst <= 1.0, f"{what}: logit error {worst:.2f} x the tolerance (rel_tol {rel_tol.tolist()})"
    v, i = ref.topk(2, dim=-1)
    clear = (v[:, 0] - v[:, 1]) > E[:, 0] * (norms[i[:, 0]] + norms[i[:, 1]])
    assert torch.equal(clear, torch.as_tensor(g[prefix + "margins"][:, step]) > 1.0), f"{what}: clearance differs from the fixture's"
    return got, clear, worst


def check_step(logits, g, step, what, norms, prefix=""):
    """Every entry within tolerance AND the greedy token exact: the fixture's lm_head seed was searched so that every step
    is clear (asserted here too)."""
    got, clear, worst = logits_vs_fixture(logits, g, step, norms, what, prefix)
    tok = torch.as_tensor(g[prefix + "tokens"][:, step]).long()
    assert clear.all(), f"{what}: fixture step is not clear"
    assert torch.equal(got.argmax(-1), tok), f"{what}: greedy token {got.argmax(-1).tolist()} != oracle {tok.tolist()}"
    return worst


def run_fixture(model, inputs, g, prefix, what, mask=None, pids=None):
    """Teacher-forced pass over a fixture (every step checked against the oracle's full logits, tokens exact), then the
    same request FREE-RUNNING through the graph-replayed greedy loop: the token matrix must equal the oracle's."""
    ref_tok = torch.as_tensor(g[prefix + "tokens"]).long()
    n = ref_tok.shape[1]
    norms = head_row_norms(model)
    logits, cache = model(**inputs, max_tokens=n)
    worst = 0.0
    for step in range(n):
        worst = max(worst, check_step(logits[:, -1], g, step, f"{what} step {step}", norms, prefix))
        if step + 1 < n:
            logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to(model.device, torch.int32), cache)
    logits, cache = model(**inputs, max_tokens=n)
    tok = model_tok(logits)
    free = [tok]
    for _ in range(n - 1):
        _, tok = model.greedy_step(tok, cache)
        free.append(tok.clone())
    assert torch.equal(torch.cat(free, dim=1).cpu().long(), ref_tok), f"{what}: free-running greedy tokens differ"
    print(f"{what}: {n} steps token-exact, worst logit error {worst:.2f} x tolerance, min clearance {g[prefix + 'margins'].min():.2f}")
    return worst


def _tiny_with_head(blind, g, key):
    from phi_3_vision_mlx_amd.api import load_synthetic
    return load_synthetic(blind_model=blind, tiny=True, seed=0, std_scale=4.0, device="cuda:0",
                          lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g[key + "head_seed"][0]))


def test_tiny_fixture_text_and_batch():
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    model, proc = _tiny_with_head(True, g, "text_")
    run_fixture(model, {"input_ids": g["text_ids"]}, g, "text_", "tiny text")
    model, proc = _tiny_with_head(True, g, "batch_")
    inputs = proc(["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"])
    run_fixture(model, inputs, g, "batch_", "tiny batch")


def test_tiny_fixture_vision():
    from golden_inputs import make_image
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    model, proc = _tiny_with_head(False, g, "vis_")
    inputs = proc("<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", [make_image(336, 336, "noise", 0)])
    assert np.asarray(inputs["input_ids"]).shape[1] == int(g["vis_n_ids"][0])
    run_fixture(model, inputs, g, "vis_", "tiny vision")


# ---------------------------------------------------------------------------------------------------------------------
# The HIP path against the REFERENCE'S OWN model / loop code: tests/golden/ref_model_tiny.npz was written by running
# /root/reference's phi.py classes and phi_3_vision_mlx.py loops over tests/golden/mlx_shim.py (gen_golden_refmodel.py);
# tests/test_refmodel.py pins the oracle to the same file on CPU.
# ---------------------------------------------------------------------------------------------------------------------
REF_IMAGES = {"sq": (336, 336, "noise", 0), "land": (640, 480, "smooth", 1)}
REF_PROMPTS = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
REF_CASES = {
    "text": (True, REF_PROMPTS[0], None), "batch": (True, REF_PROMPTS, None),
    "long": (True, "<|user|>\n" + ("the quick brown fox jumps over the lazy dog. " * 92) + "<|end|>\n<|assistant|>\n", None),
    "vis": (False, "<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", ["sq"]),
    "visns": (False, "<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", ["land"]),
    "vis2": (False, "<|user|>\n<|image_1|>\n<|image_2|>\nCompare the two.<|end|>\n<|assistant|>\n", ["sq", "land"]),
    "lora": (True, REF_PROMPTS[1], None),
    "q4cache": (True, REF_PROMPTS[0], None),        # the reference's own quantised cache (phi.py:528-540) -> cache_format="mlx4"
}


def _ref_fixture():
    import json
    with open(GOLDEN + "/ref_model_tiny.json") as f:
        return np.load(GOLDEN + "/ref_model_tiny.npz"), json.load(f)


@pytest.mark.parametrize("name", list(REF_CASES))
def test_reference_model_fixture(name, tmp_path):
    """Greedy generation of the reference's Phi3VForCausalLM / Phi3ForCausalLM (`_generate`, phi_3_vision_mlx.py:376-409) on the
    tiny checkpoint: B = 1 text, left-padded batch, a > 4096-token prompt (long RoPE factors, phi.py:492), a 336x336 image, a
    640x480 image (13 live crops), TWO images in one prompt (phi.py:400-415) and a LoRA adapter loaded by the reference's
    `_load`.  Every step's full logits row within the tiny tolerance, every token exact (teacher-forced and free-running),
    and the public `_generate` returns the reference's decoded strings."""
    from golden_inputs import make_image
    from phi_3_vision_mlx_amd import api
    g, meta = _ref_fixture()
    blind, prompt, images = REF_CASES[name]
    adapter = None
    if name == "lora":
        from phi_3_vision_mlx_amd.weights import save_adapter
        tensors = {k[len("lora_"):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("lora_model")}
        adapter = str(tmp_path / "ad")
        save_adapter(adapter, dict(meta["lora_adapter"], model_path="m", adapter_path=adapter), tensors)
    extra = dict(use_quantized_cache=True, cache_format="mlx4") if name == "q4cache" else {}
    model, proc = api.load_synthetic(blind_model=blind, tiny=True, seed=0, std_scale=4.0, device="cuda:0", adapter_path=adapter,
                                     lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g[name + "_head_seed"][0]), **extra)
    imgs = [make_image(*REF_IMAGES[i]) for i in images] if images else None
    inputs = proc(prompt, imgs) if imgs else proc(prompt)
    assert np.array_equal(np.asarray(inputs["input_ids"]), g[name + "_input_ids"])       # the reference processor's ids
    if name + "_rel_tol" in g.files:                         # a case with its own tolerance (q4cache: see gen_golden_refmodel.q4cache_case)
        view = {k_: g[name + "_" + k_] for k_ in ("rel_tol", "tokens", "logits_bf16", "margins")}
        run_fixture(model, inputs, view, "", f"reference fixture {name}")
    else:
        run_fixture(model, inputs, g, name + "_", f"reference fixture {name}")
    n = g[name + "_tokens"].shape[1]
    texts = api._generate(model, proc, prompt, imgs, max_tokens=n, verbose=False, stream=False, mute=True)
    assert texts == meta[name]["texts"], (texts, meta[name]["texts"])


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_reference_model_fixture_full_size(name):
    """The FULL-SIZE HIP path against the reference's own code, no oracle in between: `ref_model_full.npz` = the reference's `_load`
    + processors + `_generate` + `Phi3VForCausalLM` (over the MLX stand-in, tests/golden/gen_golden_refmodel.py full) on BASELINE
    config 1's 128-token prompt and on bench.py's config-2 image request (17 CLIP crops, 2531 tokens).  Every logit of every step
    within the config's tolerance, every token exact, teacher-forced and free-running."""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + "/ref_model_full.npz")
    model, proc = load_synthetic(blind_model=name == "c1", tiny=False, seed=0, device="cuda:0", lm_head_spread=float(g["spread"][0]),
                                 lm_head_seed=int(g[name + "_head_seed"][0]))
    if name == "c1":
        inp = {"input_ids": np.load(GOLDEN + "/c1_oracle.npz")["ids"]}
    else:
        inp = vqa_request(proc.img_processor, 0)
        inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    view = {"rel_tol": g[name + "_rel_tol"], "tokens": g[name + "_tokens"], "logits_bf16": g[name + "_logits_bf16"], "margins": g[name + "_margins"]}
    run_fixture(model, inp, view, "", f"reference-composed full size {name}")
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["q4text", "q4vis", "q4batch"])
def test_reference_written_4bit_checkpoint(name, tmp_path):
    """Row f4 on the GPU against the reference's own 4-bit writer / reader: the checkpoint `_quantize` wrote (rebuilt here bit for
    bit, every tensor's sha256 checked: tests/golden/q4_ckpt.py) is loaded by `api._load` -- decoder projections and lm_head stay
    4-bit (`p3v_gemv_q4` / `p3v_dequant_q4`), embeddings, ViT and projector are dequantised -- and must reproduce what the
    reference's `_load` + `_generate` computed on `QuantizedLinear` / `QuantizedEmbedding`: logits inside 4.5 %, tokens exact.
    q4batch (round 6): the two prompts as one left-padded B = 2 batch, as the reference ran them."""
    import q4_ckpt
    from golden_inputs import make_image
    from phi_3_vision_mlx_amd import api
    g, meta = _ref_fixture()
    q4_ckpt.build(str(tmp_path), meta[name], g[name + "_head_seed"][0], float(g["spread"][0]))
    model, proc = api._load(model_path=str(tmp_path), device="cuda:0")
    assert len(model.w4) == 2 * 4 + 1 and "lm_head.weight" not in model.w
    if name == "q4text":
        inputs = proc(REF_PROMPTS[1])
    elif name == "q4batch":
        inputs = proc(REF_PROMPTS)
    else:
        inputs = proc(REF_CASES["vis"][1], [make_image(*REF_IMAGES["sq"])])
    view = {"rel_tol": g[name + "_rel_tol"], "tokens": g[name + "_tokens"], "logits_bf16": g[name + "_logits_bf16"], "margins": g[name + "_margins"]}
    run_fixture(model, inputs, view, "", f"reference 4-bit checkpoint {name}")


def test_reference_choose_fixture():
    """`_choose_from` (phi_3_vision_mlx.py:466-487) under a head whose option margins are clear: the reference's picks."""
    from phi_3_vision_mlx_amd import api
    g, meta = _ref_fixture()
    model, proc = api.load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0",
                                     lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["loops_head_seed"][0]))
    assert api._choose_from(model, proc, REF_PROMPTS, "ABCDE", mute=True) == meta["loops"]["choose"]
    assert api._choose_from(model, proc, REF_PROMPTS[1], "ABCDE", mute=True) == meta["loops"]["choose_single"]
    # the reference's `_constrain` calls under the same head (plain and beam): the ids every call was fed are the loop's
    # decisions.  A score comparison of the loop is never "clear" (means of log-probabilities a few % of max|logit| apart),
    # so the HIP loop must reproduce the reference's calls up to its first differing decision and is checked decision by
    # decision against a live oracle in test_generate_choose_constrain_match_oracle_loops; here: call 0 and call 1 (prefill
    # + constraint scoring, no decision yet) feed the reference's ids, and the final text keeps the constraint.
    cons = tuple(meta["loops"]["constraint"])
    for ub in (False, True):
        ps = REF_PROMPTS if not ub else REF_PROMPTS[:1] * 2
        full = api._constrain(model, proc, list(ps), [cons], return_full_text=True, mute=True, use_beam=ub, verbose=False)
        ref = meta["loops"][f"constrain_beam{int(ub)}"]["full_text"]
        assert len(full) == len(ref) and all(t.endswith(cons[1]) for t in full)
        same = sum(a == b for a, b in zip(full, ref))
        print(f"constrain beam={ub}: {same} of {len(ref)} final texts identical to the reference's")
    # Round 5: heads under which EVERY decision of the loop is clear (searched on the oracle's decision trace, one seed for the
    # plain loop and one for the beam loop: gen_golden_refmodel.constrain_clear_case) -- there the final texts of the
    # reference's own `_constrain` are ASSERTED.
    clear = meta["loops_clear"]
    for ub in (False, True):
        info = clear[f"beam{int(ub)}"]
        assert info["min_margin"] > clear["clear_margin"]
        m2, p2 = api.load_synthetic(blind_model=True, tiny=True, seed=0, std_scale=4.0, device="cuda:0",
                                    lm_head_spread=float(g["spread"][0]), lm_head_seed=int(info["head_seed"]))
        ps = REF_PROMPTS if not ub else REF_PROMPTS[:1] * 2
        full = api._constrain(m2, p2, list(ps), [tuple(clear["constraint"])], return_full_text=True, mute=True, use_beam=ub, verbose=False)
        assert full == info["full_text"], f"constrain beam={ub} under the clear head: {full} != the reference's {info['full_text']}"


def test_parity_check_can_fail():
    """The fixture check must FAIL on a model that is subtly wrong: (a) one layer's o_proj scaled by 1.25, (b) the RoPE
    magnitude factor of a 64k instead of a 128k context window (1.16 instead of 1.19), (c) SiLU gate and up halves of one
    MLP swapped.  Each fault is injected into the tiny HIP model and the tiny text fixture must reject it."""
    g = np.load(GOLDEN + "/tiny_oracle.npz")
    model, _ = _tiny_with_head(True, g, "text_")
    inputs = {"input_ids": g["text_ids"]}
    run_fixture(model, inputs, g, "text_", "sound model")

    def rejected(what):
        try:
            run_fixture(model, inputs, g, "text_", what)
        except AssertionError:
            return True
        return False
    k = "model.layers.1.self_attn.o_proj.weight"
    good = model.w[k].clone()
    model.w[k].mul_(1.25)
    assert rejected("o_proj x1.25")
    model.w[k].copy_(good)
    model.cfg.max_position_embeddings //= 2
    assert rejected("rope magnitude")
    model.cfg.max_position_embeddings *= 2
    k = "model.layers.0.mlp.gate_up_proj.weight"
    good = model.w[k].clone()
    half = good.shape[0] // 2
    model.w[k].copy_(torch.cat([good[half:], good[:half]]))
    assert rejected("gate/up swapped")
    model.w[k].copy_(good)
    run_fixture(model, inputs, g, "text_", "restored model")


def test_synthetic_weights_identical_on_gpu_and_cpu():
    from phi_3_vision_mlx_amd.weights import peaked_lm_head, synth_values
    a, b = synth_values(300001, 12345, 0.02, device="cpu"), synth_values(300001, 12345, 0.02, device="cuda:0")
    assert torch.equal(a, b.cpu())
    w = synth_values(4096 * 64, 7, 0.02).reshape(4096, 64)
    assert torch.equal(peaked_lm_head(w, 4.0, 3), peaked_lm_head(w.cuda(), 4.0, 3).cpu())


def _full_model(g, blind=False, **kw):
    from phi_3_vision_mlx_amd.api import load_synthetic
    return load_synthetic(blind_model=blind, tiny=False, seed=0, device="cuda:0", lm_head_spread=float(g["spread"][0]),
                          lm_head_seed=int(g["head_seed"][0]), **kw)


@pytest.fixture(scope="module")
def full_text():
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, proc = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0")
    yield model, proc
    del model
    torch.cuda.empty_cache()


def test_c1_fixture_full_size():
    """BASELINE config 1 (Phi-3-mini-128K, 128-token prompt, greedy) at FULL size vs the oracle fixture generated on CPU
    (tests/golden/gen_golden_oracle.py): same hash-seeded weights on both sides; 8 steps, every vocabulary entry within
    tolerance, every greedy token exact, teacher-forced and free-running."""
    g = np.load(GOLDEN + "/c1_oracle.npz")
    model, _ = _full_model(g, blind=True)
    run_fixture(model, {"input_ids": g["ids"]}, g, "", "C1")
    del model
    torch.cuda.empty_cache()


def test_c2_fixture_full_size_vision():
    """BASELINE config 2 = bench.py's rank-0 request (one seeded 336x336 image -> 17 CLIP crops -> 2509 image tokens,
    2531-token prompt) at FULL size vs the oracle fixture: CLIP tower, projector, HD merge, 32 decoder layers, prefill +
    3 graph-replayed decode steps; token-exact on every step."""
    from golden_inputs import vqa_request
    g = np.load(GOLDEN + "/c2_oracle.npz")
    model, proc = _full_model(g)
    inp = vqa_request(proc.img_processor, 0)
    assert inp["input_ids"].shape[1] == int(g["n_ids"][0]) == 2531
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    run_fixture(model, inp, g, "", "C2")
    del model
    torch.cuda.empty_cache()


def _walk_long_fixture(model, inp, g, prefix, rel_tol, what):
    """Teacher-forced walk along a long greedy run recorded per step as (token, top-8, 256 seeded entries, max |z|, log-sum-exp,
    clearance) -- gen_golden_oracle.pack_long -- through the graph-replayed step: every recorded logit inside the tolerance at EVERY
    step, the token exact on every clear step.  Returns (exact steps, clear steps, worst error in units of the tolerance)."""
    norms = head_row_norms(model)
    toks = torch.as_tensor(g[prefix + "tokens"]).long()                      # [1, n]
    n = toks.shape[1]
    ids = torch.cat([torch.as_tensor(g[prefix + "top_ids"]).long(), torch.as_tensor(g[prefix + "sample_ids"]).long()[None, None].expand(1, n, -1)], -1)
    ref = torch.cat([_from_bits(g[prefix + "top_bf16"]), _from_bits(g[prefix + "sample_bf16"])], -1)    # [1, n, 8 + 256]
    clear = torch.as_tensor(g[prefix + "margins"]) > 1.0
    logits, cache = model(**inp, max_tokens=n)
    worst, exact, worst_lse = 0.0, 0, 0.0
    for step in range(n):
        got = logits[:, -1].float().cpu().reshape(1, -1)
        sel, r = ids[:, step], ref[:, step]
        E = rel_tol * float(g[prefix + "zmax"][0, step]) * norms[sel]
        err = ((got.gather(1, sel) - r).abs() - 2.0 ** -7 * r.abs()).clamp_min(0) / E
        worst = max(worst, err.max().item())
        worst_lse = max(worst_lse, abs(torch.logsumexp(got, -1).item() - float(g[prefix + "lse"][0, step])))
        if clear[0, step]:
            assert got.argmax(-1).item() == toks[0, step].item(), f"{what} step {step}: clear step, other token"
            exact += 1
        if step + 1 < n:
            logits, _ = model.greedy_step(toks[:, step:step + 1].to(model.device, torch.int32), cache)
    print(f"{what}: token-exact on {exact} of {int(clear.sum())} clear steps ({n} steps), worst logit error "
          f"{worst:.2f} x tolerance (rel_tol {rel_tol}), worst |log-sum-exp error| {worst_lse:.3f}")
    assert worst <= 1.0, f"{what}: logit error {worst:.2f} x the tolerance"
    return exact, int(clear.sum()), worst


def test_well_conditioned_reference_long_horizon():
    """Round 5: the end-to-end check on a network that does NOT amplify rounding noise.  Full-size text model
    with its residual-branch output projections scaled by 1 / 512 (weights.synth_weights(residual_scale=...), the fixture names the
    value: the 64 branches then carry about half the amplitude of the embedding stream; gen_golden_refmodel.wc); config 1's
    128-token prompt; 128 greedy tokens produced by the REFERENCE'S OWN `_generate` over the functional MLX stand-in
    (tests/golden/gen_golden_refmodel.py wc -> ref_model_wc.npz) under two UNSEARCHED heads -- the plain N(0, 0.02) lm_head that
    bench.py times, and the peaked head of seed 0.  The HIP path, teacher-forced through the graph-replayed step: every recorded
    logit within the fixture's rel_tol (1.2 %; measured 0.93 %) at every step, the greedy token exact on every clear step; at least
    100 of the 128 steps are clear under the unsearched peaked head and at least 85 under the plain head (tests/test_refmodel.py
    asserts the same counts on the fixture).  What this check can see: with the branches at 1 / 1024 the 64 of them carry about a
    quarter of the amplitude of the embedding stream, so a gross fault in one layer (a transposed operand, a wrong head) moves the
    logits by several per cent and fails; a subtle one (a 25 % scale error in one projection) does not -- the 2-layer fixtures at
    3 % (test_parity_check_can_fail) and the kernel tests at 2^-6 are what catch those.  tools/scratch/wc_probe.py: the HIP path's
    decode-vs-prefill self-consistency is 5.8 / 5.4 / 4.2 / 2.6 / 1.1 % of max |logit| at branch scales 1, 1/8, 1/64, 1/256, 1/1024
    (0.7 % of it is one bf16 ulp of the logits), which is why the depth-scaled 1 / sqrt(2 * 32) was not enough."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + "/ref_model_wc.npz")
    rel_tol = float(g["rel_tol"][0])
    assert rel_tol <= 0.015 + 1e-9                       # (<= 1.5 %)
    inp = {"input_ids": np.load(GOLDEN + "/c1_oracle.npz")["ids"]}
    for prefix, kw in (("plain_", {}), ("peaked0_", dict(lm_head_spread=float(g["spread"][0]), lm_head_seed=0))):
        model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", residual_scale=float(g["residual_scale"][0]), **kw)
        exact, n_clear, _ = _walk_long_fixture(model, inp, g, prefix, rel_tol, f"well-conditioned C1, {prefix[:-1]} head vs the reference")
        assert exact == n_clear
        need = 85 if prefix == "plain_" else 100
        assert n_clear >= need, f"only {n_clear} of {g[prefix + 'tokens'].shape[1]} steps are clear under the unsearched {prefix[:-1]} head"
        del model
        torch.cuda.empty_cache()


def test_well_conditioned_reference_image_path():
    """Round 6: the IMAGE path on a network that does not amplify rounding noise.  Full-size vision model, decoder
    residual branches x 1 / 1024 (as the text fixture above; the CLIP tower and the projector as they are), BASELINE config 2's request
    (bench.py's 336 x 336 image: 17 crops, 2509 image tokens + 22 text tokens); 16 greedy tokens produced by the REFERENCE'S OWN
    `_generate` over the functional MLX stand-in (tests/golden/gen_golden_refmodel.py wc_c2 -> ref_model_wc_c2.npz) under two
    UNSEARCHED heads.  The HIP path, teacher-forced through the graph-replayed step: every recorded logit within 1.2 % of max |z| at
    every step (measured 0.76 %; the plain-checkpoint C2 fixtures need 4.5 %), the greedy token exact on every clear step (24 of 32)."""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + "/ref_model_wc_c2.npz")
    rel_tol = float(g["rel_tol"][0])
    assert rel_tol <= 0.012 + 1e-9                        # (VERDICT r05 item 1c asked for <= 2 %)
    n_clear_all = 0
    for prefix, kw in (("plain_", {}), ("peaked0_", dict(lm_head_spread=float(g["spread"][0]), lm_head_seed=0))):
        model, proc = load_synthetic(blind_model=False, tiny=False, seed=0, device="cuda:0", residual_scale=float(g["residual_scale"][0]), **kw)
        inp = vqa_request(proc.img_processor, 0)
        inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
        assert np.asarray(inp["input_ids"]).shape[1] == int(g["n_ids"][0])
        exact, n_clear, _ = _walk_long_fixture(model, inp, g, prefix, rel_tol, f"well-conditioned C2 (image path), {prefix[:-1]} head vs the reference")
        assert exact == n_clear
        n_clear_all += n_clear
        del model
        torch.cuda.empty_cache()
    assert n_clear_all >= 16, f"only {n_clear_all} of 32 steps are clear under the two unsearched heads"


@pytest.mark.parametrize("name,kw5", [("c5_wc", {}), ("c5w_wc", dict(fp8_activations=False))])
def test_well_conditioned_config5_fixtures(name, kw5):
    """Round 6: config 5 -- e4m3 weights, int8 KV, e4m3 prompt activations (c5_wc: W8A8) or bf16 ones (c5w_wc:
    weight-only) -- on the WELL-CONDITIONED checkpoint against the oracle with the same three quantisers (gen_golden_oracle.c5_wc;
    the reference has no fp8 path), BASELINE config 2's request, 8 greedy steps under two unsearched heads: every recorded logit
    within 3 % (W8A8; measured 1.6 %) / 2 % (weight-only; measured 1.0 %) of max |z| -- the plain-checkpoint fixtures need 25 % / 7 %,
    because 32 amplifying layers sit behind every flipped e4m3 code -- and the token exact on every clear step."""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + f"/{name}_oracle.npz")
    rel_tol = float(g["rel_tol"][0])
    assert rel_tol <= (0.03 if name == "c5_wc" else 0.02) + 1e-9      # (VERDICT r05 item 1c asked for <= 8 % on W8A8)
    for prefix, kw in (("plain_", {}), ("peaked0_", dict(lm_head_spread=float(g["spread"][0]), lm_head_seed=0))):
        model, proc = load_synthetic(blind_model=False, tiny=False, seed=0, device="cuda:0", residual_scale=float(g["residual_scale"][0]),
                                     quantized_fp8=True, use_quantized_cache=True, **kw5, **kw)
        inp = vqa_request(proc.img_processor, 0)
        inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
        assert np.asarray(inp["input_ids"]).shape[1] == int(g["n_ids"][0])
        exact, n_clear, _ = _walk_long_fixture(model, inp, g, prefix, rel_tol, f"well-conditioned {name}, {prefix[:-1]} head vs the config-5 oracle")
        assert exact == n_clear
        del model
        torch.cuda.empty_cache()


def test_well_conditioned_4bit_fixture():
    """Round 6: MLX 4-bit group-64 decoder weights (`quantize_model=True, quantize_format="int4"`) at FULL size on the well-conditioned
    checkpoint against the oracle on the exact dequantised values (gen_golden_oracle.q4_wc), BASELINE config 1's 128-token prompt, 16
    greedy steps under two unsearched heads: every recorded logit within 2 % of max |z|, the token exact on every clear step.  The
    steps run what round 6 built for this format: k_gemv3_q4 with the step's two ends folded in (p3v_gemv_q4_step) and the merge
    launch that carries the 4-bit o_proj (k_attn_combine_o); the prefill dequantises per projection."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + "/q4_wc_oracle.npz")
    rel_tol = float(g["rel_tol"][0])
    assert rel_tol <= 0.02 + 1e-9
    for prefix, kw in (("plain_", {}), ("peaked0_", dict(lm_head_spread=float(g["spread"][0]), lm_head_seed=0))):
        model, proc = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", residual_scale=float(g["residual_scale"][0]),
                                     quantized_int4=True, **kw)
        assert len(model.w4) == 32 * 4 + 1
        inp = {"input_ids": g["ids"]}
        assert np.asarray(inp["input_ids"]).shape[1] == int(g["n_ids"][0])
        exact, n_clear, _ = _walk_long_fixture(model, inp, g, prefix, rel_tol, f"well-conditioned 4-bit weights, {prefix[:-1]} head vs the oracle")
        assert exact == n_clear and n_clear >= 8
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["c1", "c2"])
def test_long_horizon_fixtures_full_size(name):
    """Token-level parity over the BENCHMARK'S horizon: the oracle's own greedy run of BASELINE config 1 over 128
    steps (the config's 128 new tokens; the reference's benchmark() generates 100, phi_3_vision_mlx.py:1272) and of config 2
    over 32 steps, teacher-forced through the graph-replayed decode step, under TWO heads: the fixture's decisive head (its seed
    was searched for the first few steps only -- later steps are clear or not as they come) and the PLAIN N(0, 0.02) head
    bench.py times (no search, no peaking).  Every step: the top-8 entries, 256 seeded vocabulary entries and the log-sum-exp
    inside the tolerance model of the short fixtures; the greedy token exact on EVERY clear step.  Reported: k of n clear steps,
    the worst error in units of the tolerance."""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + f"/{name}_long_oracle.npz")
    rel_tol = float(g["rel_tol"][0])
    for prefix, kw in (("peaked_", dict(lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]))), ("plain_", {})):
        model, proc = load_synthetic(blind_model=name == "c1", tiny=False, seed=0, device="cuda:0", **kw)
        if name == "c1":
            inp = {"input_ids": np.load(GOLDEN + "/c1_oracle.npz")["ids"]}
        else:
            inp = vqa_request(proc.img_processor, 0)
            inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
        assert np.asarray(inp["input_ids"]).shape[1] == int(g["n_ids"][0])
        exact, n_clear, worst = _walk_long_fixture(model, inp, g, prefix, rel_tol, f"{name} {prefix[:-1]} head")
        del model
        torch.cuda.empty_cache()
    assert exact >= 0


def test_c3_long_rope_fixture_full_size():
    """BASELINE config 3's path against the oracle at a size its O(S^2) formulation can hold: a 5000-token text prompt on the
    full-size model -> S + max_tokens > 4096 -> LONG RoPE factors chosen once (phi.py:492), prefill through the big-tile
    GEMMs and the long flash prefill, 3 graph-replayed decode steps; every logit in tolerance, every token exact
    (teacher-forced and free-running).  32768 tokens stay with the property test below (the reference's own formulation
    would need 137 GB of scores per layer there)."""
    g = np.load(GOLDEN + "/c3_oracle.npz")
    model, _ = _full_model(g, blind=True)
    ids = np.random.default_rng(4).integers(3, 32000, (1, 5000)).astype(np.int64)
    assert ids.shape[1] == int(g["n_ids"][0])
    run_fixture(model, {"input_ids": ids}, g, "", "C3 (5000 tokens, long factors)")
    del model
    torch.cuda.empty_cache()


def test_c4_share_batched_vs_per_request_oracle():
    """BASELINE config 4, one GPU's share: 4 single-image VQA requests + 4 text prompts of 65..233 tokens run as ONE
    left-padded B = 8 batch (batched ViT over 68 crops, B = 8 prefill, B = 8 graph-replayed decode through the MFMA
    skinny projections and the streaming decode attention) against the PER-REQUEST B = 1 oracle runs of the fixture --
    B = 1 is the reference's only image path (phi_3_vision_mlx.py:377-378, phi.py:276), so it is the oracle of every row.
    All logits within tolerance on every row and step (rel_tol per row: image rows 4.5 %, short text rows 9 %, see the
    generator); tokens exact on every (row, step) that is clear (the head seed makes all 8 prefill steps clear; the
    fixture's count of clear decode steps is asserted, not assumed)."""
    from golden_inputs import c4_share
    from phi_3_vision_mlx_amd.processor import collate_requests
    g = np.load(GOLDEN + "/c4_oracle.npz")
    model, proc = _full_model(g)
    share = c4_share(proc.img_processor)
    assert [r["input_ids"].shape[1] for r in share] == g["n_ids"].tolist()
    batch = collate_requests(share)
    batch["pixel_values"] = torch.from_numpy(batch["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    clearance = torch.as_tensor(g["margins"])
    n = ref_tok.shape[1]
    norms = head_row_norms(model)
    logits, cache = model(**batch, max_tokens=n)
    n_exact, worst = 0, 0.0
    for step in range(n):
        got, clear, w = logits_vs_fixture(logits[:, -1], g, step, norms, f"C4 step {step}")
        worst = max(worst, w)
        assert torch.equal(got.argmax(-1)[clear], ref_tok[:, step][clear]), f"C4 step {step}"
        n_exact += int(clear.sum())
        if step + 1 < n:
            logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    assert bool((clearance[:, 0] > 1).all()) and n_exact == int((clearance > 1).sum()) and n_exact >= 12
    print(f"C4 share: {n_exact} of {clearance.numel()} (row, step) tokens exact (all clear ones), worst logit error {worst:.2f} x tolerance")
    # the same share through the length-bucketed path (dist.prefill_requests: no projection ever runs on a pad row; equal-length
    # requests are prefilled together straight into rows of a slot state, decode is one B = 8 batch): free-running greedy
    # tokens of every request equal the oracle's as long as the steps are clear
    from phi_3_vision_mlx_amd.dist import generate_requests
    dev_share = [dict(r, pixel_values=torch.from_numpy(r["pixel_values"]).to("cuda:0")) if "pixel_values" in r else r for r in share]
    toks = generate_requests(model, proc, dev_share, n, return_tokens=True)
    n_free = 0
    for r in range(len(share)):
        for step in range(n):
            if clearance[r, step] <= 1:
                break
            assert toks[r][step] == int(ref_tok[r, step]), (r, step, toks[r], ref_tok[r].tolist())
            n_free += 1
    assert n_free >= 12
    print(f"C4 share, length-bucketed prefill: {n_free} free-running tokens exact")
    del model, cache
    torch.cuda.empty_cache()


@pytest.mark.parametrize("act8", [True, False], ids=["w8a8", "w8a16"])
def test_c5_fp8_weights_int8_kv_vs_quantised_oracle(act8):
    """BASELINE config 5 on config 2's request: fp8 (e4m3, per-row scale) decoder weights AND the int8 KV cache together,
    against an oracle that applies the same quantisers (tests/golden/gen_golden_oracle.py c5): e4m3 x scale weights as exact
    fp32 products, keys / values quantised per (head, token) as the build stores them, and -- w8a8, the default path --
    e4m3 activations with one scale per token row in the prompt-sized projections (the fp8-MFMA prefill).
    Tolerances (z-space, see the generator): w8a16 7 %; w8a8 25 % -- an e4m3 code is a 6-12 % step, and wherever two correct
    implementations feed a quantiser values 0.4 % apart (bf16 vs fp32 attention output) ~5 % of the codes flip by a whole
    step; measured 18 % after 32 layers, 4-9 % after 2 (test_c5_quantisers_small_model_tight is the tight check)."""
    from golden_inputs import vqa_request
    g = np.load(GOLDEN + ("/c5_oracle.npz" if act8 else "/c5w_oracle.npz"))
    model, proc = _full_model(g, quantized_fp8=True, use_quantized_cache=True, fp8_activations=act8)
    assert model.w8 and "lm_head.weight" in model.w8 and model.fp8_act == act8
    inp = vqa_request(proc.img_processor, 0)
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    n = ref_tok.shape[1]
    norms = head_row_norms(model)
    logits, cache = model(**inp, max_tokens=n)
    assert cache[0].state.quantized
    n_exact, worst = 0, 0.0
    for step in range(n):
        got, clear, w = logits_vs_fixture(logits[:, -1], g, step, norms, f"C5 step {step}")
        worst = max(worst, w)
        assert torch.equal(got.argmax(-1)[clear], ref_tok[:, step][clear]), f"C5 step {step}"
        n_exact += int(clear.sum())
        if step + 1 < n:
            logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    assert bool(torch.as_tensor(g["margins"])[:, 0].gt(1).all()) and n_exact == int((torch.as_tensor(g["margins"]) > 1).sum()) >= 1
    print(f"C5 {'W8A8' if act8 else 'W8A16'}: {n_exact} of {n} tokens exact (the clear steps), worst logit error {worst:.2f} x tolerance")
    del model, cache
    torch.cuda.empty_cache()


HEAVY = {"c2h": {}, "c5wh": dict(quantized_fp8=True, use_quantized_cache=True, fp8_activations=False),
         "c5h": dict(quantized_fp8=True, use_quantized_cache=True, fp8_activations=True)}


@pytest.mark.parametrize("tiny", [True, False], ids=["tiny", "full"])
@pytest.mark.parametrize("tag", ["c2h", "c5wh", "c5h"])
def test_heavy_tailed_activations_fixtures(tag, tiny):
    """Config 2's request on weights with HEAVY-TAILED activations (weights.add_outliers: six residual-stream channels x 64,
    two key / value dimensions per head x 8 -- what trained decoders show and N(0, s) weights never do), in the three
    arithmetic variants of the build: bf16 (c2h), fp8 weights + int8 KV (c5wh), W8A8 prompt projections + int8 KV (c5h),
    each against an oracle with the same weights and quantisers (tests/golden/gen_golden_oracle.py `heavy`).  Every logit
    inside the fixture's tolerance (1.25-1.3 x the measured HIP - oracle difference, taken over two equally correct ViT softmax
    variants: profiles/r04_heavy_tail.txt; the double-rounded RMSNorm of round 4 raised the spread 1.3-1.6 x, a 1-ulp flip of a
    normalised outlier channel being a quarter of a typical entry), tokens exact on every clear step -- the first two steps of
    c2h / c5wh by construction; under W8A8 the per-row e4m3 activation scale is set by the outlier channels and two correct
    implementations differ by 12-24 % of the logit range per step, so c5h pins the logits only, over two steps.  The tiny
    quantised twins stop after three steps: later steps are decode steps where an ill-conditioned softmax (attention logits in
    the hundreds under 8x key dimensions) turns a single int8 / e4m3 code flip into tens of per cent -- measured and explained
    in the generator's header, not fixture material.
    (c5wh's full-size bound is 13 % since round 5, 11 % before.  Round 6 looked for a launcher setting that restores the old figure
    -- `gemm_no_skinny` + `gemm_rows=0`, the round-4 GEMM paths -- and found none: 12.1 % either way; the step comes from the rotation's
    products being rounded separately since round 5, not from a GEMM variant.  The tight bound on this arithmetic is
    test_well_conditioned_config5_fixtures: 2 % / 3 %.)"""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(f"{GOLDEN}/{'tiny_' if tiny else ''}{tag}_oracle.npz")
    model, proc = load_synthetic(tiny=tiny, seed=0, device="cuda:0", std_scale=4.0 if tiny else 1.0, outliers=True,
                                 lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]), **HEAVY[tag])
    inp = vqa_request(proc.img_processor, 0)
    assert inp["input_ids"].shape[1] == int(g["n_ids"][0])
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    ref_tok = torch.as_tensor(g["tokens"]).long()
    n = ref_tok.shape[1]
    norms = head_row_norms(model)
    logits, cache = model(**inp, max_tokens=n)
    n_exact, worst = 0, 0.0
    for step in range(n):
        got, clear, w = logits_vs_fixture(logits[:, -1], g, step, norms, f"{tag} step {step}")
        worst = max(worst, w)
        assert torch.equal(got.argmax(-1)[clear], ref_tok[:, step][clear]), f"{tag} step {step}"
        n_exact += int(clear.sum())
        if step + 1 < n:
            logits, _ = model.greedy_step(ref_tok[:, step:step + 1].to("cuda:0", torch.int32), cache)
    assert n_exact == int((torch.as_tensor(g["margins"]) > 1).sum())
    if tag != "c5h":
        assert n_exact >= 2
    print(f"heavy tails, {'tiny' if tiny else 'full size'} {tag}: worst logit error {worst:.2f} x tolerance ({float(g['rel_tol'][0]):.3f}), "
          f"{n_exact} of {n} tokens pinned and exact")
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
    # two correct orders of the same bf16 arithmetic (decode kernels on a 32768-row cache vs one 32769-row prefill) differ by
    # 5.2 % of max|logit| on this random network (tools/c3_tol_probe.py, round 3): the bar cannot be tighter than that
    assert_logits(b[:, -1], c[:, -1], "32k decode vs prefill", rel_atol=6e-2)


def test_full_size_decode_equals_prefill_property(full_text):
    """Size-independent property at full size: (prefill S, decode 1) == (prefill S+1) on the last row.
    Why 6 % and not the 2 % SURVEY.md App. A suggests: the two sides are two correct ORDERS of the same bf16 arithmetic (GEMV +
    split-KV decode kernels vs GEMM + flash prefill), i.e. they differ by a handful of flipped bf16 roundings per layer, and
    profiles/r03_precision_decomposition_c1_oracle.txt measures what ONE such rounding does on this random 32-layer network
    at a short context: 5.8-6.5 % of max|logit| at the logits whichever rounding is injected (q, k, P or the attention output
    alone; residual stream 0.7 % after layer 0 -> 5.9 % after layer 31).  2 layers (the tiny model) stay under 1.5 %."""
    model, _ = full_text
    ids = rand_ids(300, 9)
    a, cache = model(input_ids=ids[:, :299], max_tokens=2)
    b, _ = model(input_ids=ids[:, 299:], cache=cache)
    c, _ = model(input_ids=ids, max_tokens=1)
    assert_logits(b[:, -1], c[:, -1], "full-size decode vs prefill", rel_atol=6e-2)   # 32 layers: see _check_topk


@pytest.mark.parametrize("B", [12, 20, 64])
def test_full_width_batched_decode_rows_equal_their_solo_runs(B):
    """Decode batches of 9+ rows run their projections on the weight-streaming GEMM (64-row tiles up to 64 rows; K slices + the
    reduction that also normalises), B = 1 on the GEMV kernels: on a full-WIDTH 2-layer model every row of a graph-replayed batch
    step must match that row decoded alone -- same arithmetic, another kernel family and summation order."""
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", num_hidden_layers=2)
    ids = np.random.default_rng(B).integers(3, 32000, (B, 40)).astype(np.int64)
    lg, cache = model(input_ids=ids, max_tokens=4)
    tok = ops.argmax(lg[:, -1].contiguous())[:, None]
    batch = [lg[:, -1].float().cpu()]
    toks = [tok.cpu()]
    for _ in range(3):
        lg, tok = model.greedy_step(tok, cache)
        batch.append(lg[:, -1].float().cpu())
        toks.append(tok.cpu())
    for r in sorted({0, B // 2, B - 1}):
        lg1, c1 = model(input_ids=ids[r:r + 1], max_tokens=4)
        assert_logits(batch[0][r:r + 1], lg1[:, -1], f"B={B} row {r} prefill")
        for step in range(3):
            lg1, _ = model.greedy_step(toks[step][r:r + 1].to("cuda:0"), c1)       # teacher-forced with the batch's tokens
            assert_logits(batch[step + 1][r:r + 1], lg1[:, -1], f"B={B} row {r} step {step}")
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [2, 7, 16])
def test_full_width_4bit_batched_decode_rows_equal_their_solo_runs(B):
    """Row f4 at M > 1 (round 6): MLX 4-bit group-64 weights (`quantize_model=True`) under a LEFT-PADDED decode batch of 2 .. 16 rows run
    k_gemm_rows_q4 -- the packed weights dequantised in registers in front of the fp16 MFMA -- where rounds 3-5 dequantised whole
    matrices into a bf16 scratch; B = 1 runs the 4-bit GEMV.  Full-width 2-layer model, rows of different lengths: every valid row of
    a graph-replayed batch step must match that row decoded alone (same weights, another kernel family), and the old path
    (P3V_Q4_ROWS=0) must agree too."""
    import os
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, _ = load_synthetic(blind_model=True, tiny=False, seed=0, device="cuda:0", num_hidden_layers=2, quantized_int4=True)
    assert len(model.w4) == 2 * 4 + 1
    S = 40
    rng = np.random.default_rng(B)
    ids = rng.integers(3, 32000, (B, S)).astype(np.int64)
    lens = [S] + [int(v) for v in rng.integers(S // 2, S, B - 1)]
    mask = np.zeros((B, S), dtype=np.int64)
    pids = np.ones((B, S), dtype=np.int64)                         # left padding as Phi3FProcessor._tokenize builds it (phi.py:233-245)
    for r, n in enumerate(lens):
        ids[r, :S - n] = 0
        mask[r, S - n:] = 1
        pids[r, S - n:] = np.arange(n)
    runs = {}
    for mode in ("1", "0"):
        os.environ["P3V_Q4_ROWS"] = mode
        try:
            lg, cache = model(input_ids=ids, pids=pids, mask=mask, max_tokens=4)
            tok = ops.argmax(lg[:, -1].contiguous())[:, None]
            batch, toks = [lg[:, -1].float().cpu()], [tok.cpu()]
            for step in range(3):
                feed = tok if mode == "1" else runs["1"][1][step].to("cuda:0")     # (teacher-forced with the first run's tokens)
                lg, tok = model.greedy_step(feed, cache)
                batch.append(lg[:, -1].float().cpu())
                toks.append(tok.cpu())
        finally:
            os.environ.pop("P3V_Q4_ROWS", None)
        runs[mode] = (batch, toks)
        del cache
    batch, toks = runs["1"]
    for step in range(4):                                          # the register-dequantising kernel against the bf16-scratch path
        assert_logits(batch[step], runs["0"][0][step].to("cuda:0"), f"4-bit B={B} step {step}: rows kernel vs dequantise + bf16 kernels", rel_atol=6e-2)
    for r in sorted({0, B // 2, B - 1}):
        n = lens[r]
        lg1, c1 = model(input_ids=ids[r:r + 1, S - n:], max_tokens=4)
        assert_logits(batch[0][r:r + 1], lg1[:, -1], f"4-bit B={B} row {r} prefill", rel_atol=6e-2)
        for step in range(3):
            lg1, _ = model.greedy_step(toks[step][r:r + 1].to("cuda:0"), c1)
            assert_logits(batch[step + 1][r:r + 1], lg1[:, -1], f"4-bit B={B} row {r} step {step}", rel_atol=6e-2)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layers,serving", [(3, False), (2, False), (2, True)], ids=["odd-stack", "even-stack", "even-stack-server"])
def test_fused_oproj_decode_needs_an_even_stack_and_an_exclusive_gpu(layers, serving, monkeypatch):
    """the fused attention + o_proj launch of layer i re-arms the OTHER parity's output buffer, so an odd number of
    layers would hand layer 0 of the next step a buffer still holding the last layer's words; and it needs every workgroup of its
    grid resident at once, which a server-owned model cannot promise.  Full-width truncations (3 and 2 layers), a context inside
    the fused range, graph-replayed steps against eager steps with the fusion switched off: bit-identical tokens and logits, the
    odd stack and the server-owned model without the fused launch."""
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    model, _ = load_synthetic(blind_model=True, seed=3, num_hidden_layers=layers, device="cuda:0")
    model.serving = serving
    ids = rand_ids(1800, 21)
    l0, cache = model(input_ids=ids, max_tokens=12)
    tok = ops.argmax(l0[:, -1, :].contiguous())[:, None]
    got = []
    t = tok
    for _ in range(8):                                       # graph replays (the second step onwards would read the stale buffer)
        lg, t = model.greedy_step(t, cache)
        got.append((lg.clone(), t.clone()))
    fused = cache[0].state.graphs["greedy"]["bufs"]["fuse_o"]
    assert fused == (layers % 2 == 0 and not serving)
    monkeypatch.setenv("P3V_ATTN_FUSE_OPROJ", "0")
    _, cache2 = model(input_ids=ids, max_tokens=12)
    t = tok
    for i in range(8):                                       # eager steps, separate o_proj launch
        lg, cache2 = model(input_ids=t, cache=cache2)
        t = ops.argmax(lg[:, -1, :].contiguous())[:, None]
        assert torch.equal(lg.view(-1), got[i][0].view(-1)), f"step {i}: logits differ"
        assert torch.equal(t.view(-1), got[i][1].view(-1)), f"step {i}: token differs"
    assert torch.isfinite(got[-1][0].float()).all()


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


@pytest.mark.parametrize("path,prefill_bound", [("tiles128", 0.045), ("default", 0.065)])
def test_int4_weights_vs_live_oracle_on_mlx_dequantised_weights(path, prefill_bound):
    """(the prefill bound went 0.045 -> 0.065 when the 160-row GEMMs moved to the 128 x 64 weight-streaming tiles; the old
    bound is still asserted on the path it was measured on -- `gemm_no_skinny` pins the 128 x 128 tiles -- and the new one on the
    launcher's own choice, with the measured values printed so drift is visible.)
    The reference's `quantize_model=True` format (4-bit group-64, `nn.quantize`, phi_3_vision_mlx.py:264,297-305) against
    the ORACLE, not against the HIP path itself: a 2-layer model of the full width (H = 3072, I = 8192, V = 32064 -- the
    shapes `p3v_gemv_q4` takes) built from MLX-format tensors (`Q4Weight`: packed codes + bf16 scales / biases, the same
    objects an MLX `*_Q` checkpoint loads into), vs the oracle on `mx.dequantize`'s arithmetic -- scale * q + bias in fp32,
    which is what MLX's quantised matmul accumulates.  Prefill runs dequantise -> bf16 MFMA (one extra bf16 rounding per
    weight), decode runs the 4-bit GEMV (exact products): every logit inside the 2-layer tolerance at every step, tokens
    equal wherever the oracle's margin is clear."""
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd.config import make_config, phi3v_config_dict
    from phi_3_vision_mlx_amd.model import Phi3VModel
    from phi_3_vision_mlx_amd.weights import Q4Weight, mlx_dequantize, mlx_quantize, synth_weights
    d = phi3v_config_dict(vision=False)
    d.update(num_hidden_layers=2)
    cfg = make_config(d)
    w = synth_weights(cfg, seed=3, lm_head_spread=4.0, lm_head_seed=1)
    wq, wo = dict(w), dict(w)
    for k in list(w):
        if k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith("_proj.weight")):
            q = mlx_quantize(w[k])
            wq[k] = Q4Weight(*q)                                       # what `load()` builds from an MLX 4-bit checkpoint
            wo[k] = mlx_dequantize(*q)                                 # fp32 scale * q + bias
    model = Phi3VModel(cfg, wq, device="cuda:0")
    assert len(model.w4) == 2 * 4 + 1 and "lm_head.weight" not in model.w
    oracle = orc.OraclePhi3V(cfg, wo, cache_fp32=True)
    ids = np.random.default_rng(8).integers(3, 32000, (1, 160)).astype(np.int64)
    n = 5
    ref, oc = oracle(input_ids=ids, max_tokens=n)
    from phi_3_vision_mlx_amd import ops
    old = ops.set_tuning("gemm_no_skinny", int(path == "tiles128"))
    try:
        got, cache = model(input_ids=ids, max_tokens=n)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("gemm_no_skinny", old)
    worst, n_tok = [], 0
    for step in range(n):
        r, gl = ref[:, -1].float(), got[:, -1].float().cpu()
        worst.append(((gl - r).abs().amax() / r.abs().amax()).item())
        v, i = r.topk(2, dim=-1)
        if (v[0, 0] - v[0, 1]) > 2 * (prefill_bound if step == 0 else 0.03) * r.abs().amax():     # clear under the tolerance asserted below
            assert int(gl.argmax(-1)) == int(i[0, 0]), (step, int(gl.argmax(-1)), int(i[0, 0]))
            n_tok += 1
        if step + 1 < n:
            tok = torch.argmax(r, dim=-1)[:, None]
            ref, oc = oracle(input_ids=tok, cache=oc)
            got, _ = model.greedy_step(tok.to("cuda:0", torch.int32), cache)
    print(f"int4 vs oracle [{path}]: worst logit error per step (fraction of max|logit|):", [round(x, 4) for x in worst], "tokens pinned:", n_tok)
    # measured 0.032-0.053 (prefill: every weight rounded once more, to bf16; the worst of 32064 logits moves that much with the
    # fp32 summation order of the 160-row GEMMs alone: 0.041 on the 128 x 128 tiles, 0.050 / 0.053 on the 128 x 64 ones in one pass /
    # in K slices) and 0.011-0.024 (decode: exact 4-bit products)
    assert worst[0] <= prefill_bound and max(worst[1:]) <= 0.03 and n_tok >= 3, (path, worst, n_tok)
    del model, cache
    torch.cuda.empty_cache()


def test_c5_quantisers_small_model_tight():
    """Config 5's three quantisers (e4m3 weights, e4m3 activations in the prompt-sized projections on the fp8 MFMA, int8 KV)
    on a 2-layer model whose shapes the fp8 GEMM takes (H = 384, I = 512): HIP vs a LIVE oracle that applies the same
    quantisers (tests/golden/gen_golden_oracle.py: c5_quantisers / c5_proj / QuantKVCache).  Two layers leave little room for
    the rounding noise to grow, so this is the tightest end-to-end check that both sides implement the SAME quantised
    arithmetic (the kernels themselves are held to bf16 rounding against fp32 math in tests/test_kernels_gpu.py)."""
    import phi3v_oracle as orc
    import gen_golden_oracle as gg
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    from phi_3_vision_mlx_amd.model import Phi3VModel
    from phi_3_vision_mlx_amd.weights import synth_weights
    d = tiny_config_dict(vision=False)
    d.update(hidden_size=384, num_attention_heads=4, num_key_value_heads=4, intermediate_size=512, quantized_fp8=True,
             use_quantized_cache=True)
    cfg = make_config(d)
    w = synth_weights(cfg, seed=0, std_scale=2.0)
    model = Phi3VModel(cfg, w, device="cuda:0")
    assert model.fp8_act and len(model.w8) == 2 * 4 + 1
    ow = gg.c5_quantisers(cfg, w)
    w8, sc = model.w8["lm_head.weight"]
    ow["lm_head.weight"] = w8.view(torch.float8_e4m3fn).float().cpu() * sc.cpu()[:, None]
    oracle = orc.OraclePhi3V(cfg, ow, cache_fp32=True)
    oracle.proj = gg.c5_proj(oracle)
    keep = orc.OracleKVCache
    orc.OracleKVCache = gg.QuantKVCache
    try:
        ids = np.random.default_rng(8).integers(3, 32000, (2, 300)).astype(np.int64)
        n = 5
        ref, oc = oracle(input_ids=ids, max_tokens=n)
        got, cache = model(input_ids=ids, max_tokens=n)
        worst = []
        for step in range(n):
            r, gl = ref[:, -1].float(), got[:, -1].float().cpu()
            worst.append(((gl - r).abs().amax() / r.abs().amax()).item())
            if step + 1 < n:
                tok = torch.argmax(r, dim=-1)[:, None]
                ref, oc = oracle(input_ids=tok, cache=oc)
                got, _ = model.greedy_step(tok.to("cuda:0", torch.int32), cache)
    finally:
        orc.OracleKVCache = keep
    print("C5 small model: worst logit error per step (fraction of max|logit|):", [round(x, 4) for x in worst])
    assert max(worst) <= 0.13, worst                            # measured 0.04-0.09: e4m3 code flips (see the C5 fixture test)
