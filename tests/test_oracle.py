"""The CPU oracle: pinned against what the reference provides for this path
(RoPE constants from the reference's notebook, token-count formula, the
committed regression fixture) and checked for the invariants the reference's
cache / mask / beam semantics imply."""
import math
import os

import numpy as np
import pytest
import torch

import phi3v_oracle as orc
from phi_3_vision_mlx_amd.config import LONG_FACTOR, SHORT_FACTOR, make_config, rope_scaling_factor, tiny_config_dict
from phi_3_vision_mlx_amd.processor import Phi3FProcessor
from phi_3_vision_mlx_amd.weights import synth_weights

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_oracle.npz"))


@pytest.fixture(scope="module")
def tiny():
    cfg = make_config(tiny_config_dict(vision=False))
    return cfg, orc.OraclePhi3V(cfg, synth_weights(cfg, seed=0, std_scale=4.0), cache_fp32=True), Phi3FProcessor(None)


def test_su_rope_tables_against_reference_constants():
    """scaling factor 1.1902380714238083 and the factor arrays are the values printed in the reference's
    assets/su_rope_explained.ipynb (cells 7, 9); short factors iff L_all <= 4096 (phi.py:492)."""
    cfg = make_config()
    assert abs(rope_scaling_factor(cfg) - 1.1902380714238083) < 1e-15
    for L_all, fac in ((4096, SHORT_FACTOR), (4097, LONG_FACTOR), (32768 + 128, LONG_FACTOR)):
        cos, sin = orc.su_rope_tables(cfg, L_all, None)
        assert cos.shape == (1, 1, L_all, 96) and cos.dtype == torch.float32
        for pos in (0, 1, 4095, min(4096, L_all - 1), L_all - 1):
            inv = 1.0 / (np.asarray(fac, dtype=np.float64) * 10000.0 ** (np.arange(0, 96, 2) / 96))
            e = pos * inv
            # fp32 angle pos*inv_freq carries ~pos*2^-24 rad of rounding (the reference computes it in fp32 too)
            assert np.allclose(cos[0, 0, pos, :48].numpy(), np.cos(e) * 1.1902380714238083, atol=2e-5 + pos * 3e-7)
            assert torch.equal(cos[0, 0, pos, :48], cos[0, 0, pos, 48:])       # emb = cat(freqs, freqs)
    # left-pad position ids are extended by +1 per generated token (phi.py:496-497)
    pids = torch.tensor([[1, 1, 0, 1, 2]])
    c2, _ = orc.su_rope_tables(cfg, 8, pids)
    c_ref, _ = orc.su_rope_tables(cfg, 8, None)
    assert torch.equal(c2[0, 0, 5:], c_ref[0, 0, 3:6]) and torch.equal(c2[0, 0, 2], c_ref[0, 0, 0])


def test_rotate_half_is_half_split_convention():
    x = torch.arange(8, dtype=torch.float32).reshape(1, 1, 1, 8).to(torch.bfloat16)
    cos, sin = torch.zeros(1, 1, 1, 8), torch.ones(1, 1, 1, 8)
    out = orc.rotate_half(x, cos, sin)                      # pure rotation by 90 degrees: (-x2, x1)
    assert out.dtype == torch.float32 and out.flatten().tolist() == [-4, -5, -6, -7, 0, 1, 2, 3]


def _from_bits(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()


def test_regression_fixture_tiny(tiny):
    """The committed fixture (tests/golden/gen_golden_oracle.py) still reproduces (tokens exactly, logits to one bf16 ulp): greedy tokens and the FULL
    last-position logits of every step under the fixture's decisive lm_head, choose, constrain; and every greedy step of
    the fixture is clear (top-2 margin > the sum of the two entries' tolerances), so the GPU tests assert exact token ids."""
    from phi_3_vision_mlx_amd.weights import peaked_lm_head
    cfg, o, proc = tiny
    base = o.w["lm_head.weight"]
    spread, rel_tol = float(GOLD["spread"][0]), float(GOLD["rel_tol"][0])

    def with_head(hs):
        o.w["lm_head.weight"] = peaked_lm_head(base, spread, int(hs))
        o._f32.pop("lm_head.weight", None)
    try:
        prompts = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
        inp = proc(prompts)
        for key, inputs in (("text_", {"input_ids": GOLD["text_ids"]}), ("batch_", inp)):
            with_head(GOLD[key + "head_seed"][0])
            n = GOLD[key + "tokens"].shape[1]
            toks, lgs = orc.greedy_generate(o, dict(inputs), n, stop_on_eos=False)
            assert np.array_equal(toks.numpy(), GOLD[key + "tokens"])
            ref = _from_bits(GOLD[key + "logits_bf16"])                 # the fixture projected the last row only: a different
            # GEMM blocking may flip a last bf16 bit; entries that cancel to ~0 carry the fp32 sum's absolute noise instead
            assert (lgs.float() - ref).abs().le(2.0 ** -7 * ref.abs() + 2.0 ** -18 * ref.abs().max()).all()
            from gen_golden_oracle import clearance, row_norms        # the fixture's tolerance model (row-normalised logits)
            cl = clearance(lgs, row_norms(o.w["lm_head.weight"]), rel_tol)
            assert np.allclose(cl.numpy(), GOLD[key + "margins"], rtol=0.1) and cl.min().item() > 1.0
        with_head(GOLD["choose_head_seed"][0])
        opts = proc([f" {c}" for c in "ABCDE"])["input_ids"][:, -1]
        assert orc.choose_from(o, proc(prompts), opts) == GOLD["choose_idx"].tolist()
        idc = proc.tokenizer.encode(" The answer is", add_special_tokens=False)[1:]
        for ub in (0, 1):
            cin = dict(inp) if not ub else proc(prompts[:1] * 2)
            trace = []
            s, sc = orc.constrain_one(o, cin, (3, " The answer is"), idc, use_beam=bool(ub), trace=trace)
            assert np.array_equal(s.numpy(), GOLD[f"constrain_beam{ub}_synth"])
            assert np.allclose(sc.float().numpy(), GOLD[f"constrain_beam{ub}_score"])
            assert len(trace) == (7 if not ub else 19) and all(m >= 0 for _, m, _ in trace)
    finally:
        o.w["lm_head.weight"] = base
        o._f32.pop("lm_head.weight", None)


def test_cache_semantics(tiny):
    cfg, o, _ = tiny
    ids = np.random.default_rng(5).integers(3, 32000, (1, 30))
    full, _ = o(input_ids=ids, max_tokens=1)
    a, cache = o(input_ids=ids[:, :29], max_tokens=4)
    b, cache = o(input_ids=ids[:, 29:], cache=cache)
    assert cache[0].offset == 30
    assert (b[:, -1].float() - full[:, -1].float()).abs().max() < 0.02           # decode == prefill (KVCache append)
    # advance_offset=0 rewinds: the same call repeated gives the same logits (phi.py:589-591)
    c1, cache = o(input_ids=ids[:, :3], cache=cache, advance_offset=0)
    c2, cache = o(input_ids=ids[:, :3], cache=cache, advance_offset=0)
    assert cache[0].offset == 30 and torch.equal(c1, c2)
    # advance_offset=1 commits only the first of the L tokens
    _, cache = o(input_ids=ids[:, :3], cache=cache, advance_offset=1)
    assert cache[0].offset == 31
    # max_tokens=0: no cache reuse, logits equal the cached prefill (choose path, phi.py:521-522)
    z, zc = o(input_ids=ids, max_tokens=0)
    assert torch.equal(z, full) and zc[0].kv is None


def test_beam_view_equals_expanded_batch(tiny):
    """n_beam: read-only repeated cache + new keys (phi.py:523-527) == running each beam as its own sequence."""
    cfg, o, proc = tiny
    inp = proc(["twelve tokens..", "also twelve tok"])          # equal lengths: no padding, but pids/mask present as in
    ids = np.asarray(inp["input_ids"])                          # the reference's _constrain (B>1 beams need per-row pids)
    S = ids.shape[1]
    _, cache = o(**inp, max_tokens=8)
    beam = np.random.default_rng(7).integers(3, 32000, (6, 3))
    got, _ = o(input_ids=beam, cache=cache, n_beam=3, advance_offset=0)
    assert cache[0].offset == S
    for r in range(6):
        solo_ids = np.concatenate([ids[r // 3], beam[r]])[None]
        ref, _ = o(input_ids=solo_ids, max_tokens=0)
        assert (got[r].float() - ref[0, -3:].float()).abs().max() < 0.03


def test_left_pad_invariance_and_q7(tiny):
    cfg, o, proc = tiny
    inp = proc(["a", "a considerably longer prompt than the first"])
    lg, _ = o(**inp, max_tokens=2)
    assert not torch.isnan(lg.float()).any()                                   # Q7: pad rows are defined (zeros), never NaN
    solo, _ = o(**proc("a considerably longer prompt than the first"), max_tokens=2)
    assert torch.equal(lg[1, -1], solo[0, -1])
    short, _ = o(**proc("a"), max_tokens=2)
    assert (lg[0, -1].float() - short[0, -1].float()).abs().max() < 0.07       # <= 2 bf16 ulp at |logit| ~ 4


def test_hd_token_count_formula():
    """(h*w+1)*144 + 1 + (h+1)*12 (phi.py:319,411) for every crop grid the HD transform can produce."""
    for h in range(1, 17):
        for w in range(1, 17):
            if h * w <= 16:
                sub = h * 12 * (w * 12 + 1)
                assert sub + 1 + 12 * 13 == (h * w + 1) * 144 + 1 + (h + 1) * 12
