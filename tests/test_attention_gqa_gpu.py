"""Grouped-query attention (n_kv < n_heads) through every attention entry point of the C ABI, against the oracle's fp32
attention with the K / V heads repeated (query head h reads kv head h // (n_heads / n_kv): HF `repeat_kv`; the
reference's Phi3Attention carries num_key_value_heads in its split, phi.py:430-446, and SURVEY.md:42-44 asks for
GQA-capable kernels).  Phi-3-mini / Phi-3-Vision ship n_kv == n_heads, which is what every other attention test uses;
these cases use 8/4 and 32/8 at head dim 96 (+ 4/1: multi-query)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
GQA = [(8, 4), (32, 8), (4, 1)]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from phi_3_vision_mlx_amd import ops as o
    return o


@pytest.fixture(scope="module")
def orc():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import phi3v_oracle
    return phi3v_oracle


def g(shape, seed, std=1.0, dtype=BF16):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=gen) * std).to(dtype)


def close(a, b, rtol=2 ** -6, atol=2e-2):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert (err <= tol).all(), f"max err {err.max().item():.5f} (worst excess {(err - tol).max().item():.5f})"


def rep_kv(t, nh):
    """[B, nkv, T, hd] -> [B, nh, T, hd], query head h reading kv head h // (nh / nkv)."""
    return t.repeat_interleave(nh // t.shape[1], dim=1)


def attn_ref(orc, q, k, v, scale, allowed):
    nh = q.shape[1]
    w = (q.float() * scale) @ rep_kv(k, nh).float().transpose(-1, -2)
    return orc.masked_softmax(w, allowed) @ rep_kv(v, nh).float()


def allowed_mask(B, L, past, pad, causal=True):
    T = past + L
    t = torch.arange(T)[None, None, None, :]
    qpos = (past + torch.arange(L))[None, None, :, None]
    a = (t >= pad[:, None, None, None]) & (qpos >= pad[:, None, None, None])
    if causal:
        a = a & (t <= qpos)
    return a.expand(B, 1, L, T)


@pytest.mark.parametrize("nh,nkv", GQA)
@pytest.mark.parametrize("B,L,past,pads", [(1, 130, 0, None), (2, 70, 0, [0, 9]), (1, 1, 300, None), (3, 1, 77, [0, 5, 70]),
                                           (2, 6, 130, [3, 0]), (1, 16, 64, None), (1, 33, 100, None)])
def test_attention_gqa(ops, orc, nh, nkv, B, L, past, pads):
    """p3v_attention (launcher's own kernel choice: prompt kernels, cached multi-row calls, split-KV with the merge launch)."""
    hd, T = 96, past + L
    Tp = (T + 63) // 64 * 64
    q, k, v = g((B, nh, L, hd), 40), g((B, nkv, T, hd), 41), g((B, nkv, T, hd), 42)
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16).cuda()
    kc, vc = torch.zeros((B, nkv, Tp, hd), dtype=BF16), torch.zeros((B, nkv, hd, Tp), dtype=BF16)
    kc[:, :, :T], vc[:, :, :, :T] = k, v.transpose(2, 3)
    n_split, ws = 0, None
    if L <= 16:
        n_split = 4
        ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    ops.attention(q.cuda(), out, B, L, nh, nkv, hd, hd ** -0.5, True, past=past, k_past=kc.cuda(), v_past=vc.cuda(), past_t=Tp,
                  pad_len=pad.cuda() if pads else None, ws=ws, n_split=n_split, new_is_cache=True)
    ref = attn_ref(orc, q, k, v, hd ** -0.5, allowed_mask(B, L, past, pad)).transpose(1, 2).reshape(B, L, nh * hd)
    valid = ((past + torch.arange(L))[None, :] >= pad[:, None])[..., None].expand(B, L, nh * hd)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    close(got[valid], ref[valid])


@pytest.mark.parametrize("pp", [2, 1, 0], ids=["interleaved", "pingpong", "dma"])
@pytest.mark.parametrize("nh,nkv", GQA)
@pytest.mark.parametrize("B,L,past,pads", [(1, 700, 0, None), (2, 300, 0, [0, 77]), (1, 40, 200, None), (1, 1100, 0, None)])
def test_attention_prefill_kernels_gqa(ops, orc, pp, nh, nkv, B, L, past, pads):
    """Each prompt-sized kernel pinned (k_attn_prefill_il with 8 and 4 waves, _pp, _dma), pre-scaled queries."""
    hd, T = 96, past + L
    Tp = (T + 63) // 64 * 64
    scale = hd ** -0.5
    q, k, v = g((B, nh, L, hd), 50), g((B, nkv, T, hd), 51), g((B, nkv, T, hd), 52)
    q_in = (q.float() * (scale * ops.Q_PRESCALE)).to(BF16)
    q_ref = q_in.float() / (scale * ops.Q_PRESCALE)
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    kc, vc = torch.zeros((B, nkv, Tp, hd), dtype=BF16), torch.zeros((B, nkv, hd, Tp), dtype=BF16)
    kc[:, :, :T], vc[:, :, :, :T] = k, v.transpose(2, 3)
    ref = attn_ref(orc, q_ref, k, v, scale, allowed_mask(B, L, past, pad)).transpose(1, 2).reshape(B, L, nh * hd)
    valid = ((past + torch.arange(L))[None, :] >= pad[:, None])[..., None].expand(B, L, nh * hd)
    for waves in ((8, 4) if pp == 2 else (8,)):
        out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16).cuda()
        saved = [ops.set_tuning("attn_pp", min(pp, 1)), ops.set_tuning("attn_il", int(pp == 2)), ops.set_tuning("attn_il_waves", waves)]
        try:
            ops.attention(q_in.cuda(), out, B, L, nh, nkv, hd, scale, True, past=past, k_past=kc.cuda(), v_past=vc.cuda(), past_t=Tp,
                          pad_len=pad.cuda() if pads else None, new_is_cache=True, q_prescaled=True)
            torch.cuda.synchronize()
        finally:
            for n, val in zip(("attn_pp", "attn_il", "attn_il_waves"), saved):
                ops.set_tuning(n, val)
        got = out.float().cpu()
        assert torch.isfinite(got).all()
        assert (got[~valid] == 0).all()
        close(got[valid], ref[valid])


def _rope(ops, orc, B, T, hd):
    from phi_3_vision_mlx_amd.config import make_config, rope_scaling_factor
    cfg = make_config()
    cos_ref, sin_ref = orc.su_rope_tables(cfg, T, None)
    inv = 1.0 / (torch.tensor(cfg.rope_scaling["short_factor"], dtype=F32) * (10000.0 ** (torch.arange(0, hd, 2, dtype=F32) / hd)))
    cos, sin = ops.rope_table(torch.arange(T, dtype=F32).repeat(B).cuda(), inv.cuda(), rope_scaling_factor(cfg))
    return cos.view(B, T, -1), sin.view(B, T, -1), cos_ref, sin_ref


@pytest.mark.parametrize("nh,nkv", GQA)
@pytest.mark.parametrize("B,L,past,n_split,pads", [(1, 1, 300, 5, None), (2, 1, 63, 1, [0, 7]), (1, 6, 130, 3, None),
                                                   (3, 1, 2000, 32, [0, 100, 1999]), (2, 4, 61, 2, [0, 7]), (1, 3, 700, 3, None)])
@pytest.mark.parametrize("dev_past,fused_merge", [(True, True), (False, False)])
@pytest.mark.parametrize("tile", [64, 128])
def test_attention_decode_fused_gqa(ops, orc, nh, nkv, B, L, past, n_split, pads, dev_past, fused_merge, tile):
    """p3v_attention_decode: head split of a [nh + 2 nkv] x hd projection row, RoPE, append into the nkv-head cache, split-KV
    attention, merge (in the launch or by the merge launch), on the 64-key, 128-key and streaming plans."""
    hd = 96
    T = (past + L + 5 + tile - 1) // tile * tile
    if tile == 128:
        n_split = T // 128
    qkv = g((B * L, (nh + 2 * nkv) * hd), 45)
    kc, vc = g((B, nkv, T, hd), 46), g((B, nkv, T, hd), 47)
    cos, sin, cos_ref, sin_ref = _rope(ops, orc, B, T, hd)
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    kcc, vcc = kc.cuda(), vc.transpose(2, 3).contiguous().cuda()
    out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ops.attention_decode(qkv.cuda(), cos[:, past:].contiguous(), sin[:, past:].contiguous(), T - past, kcc, vcc, out, B, L, nh, nkv, hd,
                         hd ** -0.5, 0 if dev_past else past, T, ws, n_split, pad_len=pad.cuda() if pads else None,
                         d_past=d_past if dev_past else None, merge_in_launch=fused_merge)
    assert (ws.view(torch.int32) == -1).all()
    x = qkv.view(B, L, nh + 2 * nkv, hd).transpose(1, 2)
    cs, sn = cos_ref[:, :, past:past + L], sin_ref[:, :, past:past + L]
    q = orc.rotate_half(x[:, :nh], cs, sn).to(BF16)
    k_new = orc.rotate_half(x[:, nh:nh + nkv], cs, sn).to(BF16)
    v_new = x[:, nh + nkv:]
    kf, vf = torch.cat([kc[:, :, :past], k_new], dim=2), torch.cat([vc[:, :, :past], v_new], dim=2)
    ref = attn_ref(orc, q, kf, vf, hd ** -0.5, allowed_mask(B, L, past, pad)).transpose(1, 2).reshape(B, L, nh * hd)
    valid = ((past + torch.arange(L))[None, :] >= pad[:, None])[..., None].expand(B, L, nh * hd)
    got = out.float().cpu()
    assert torch.isfinite(got[valid]).all()
    close(got[valid], ref[valid])
    close(kcc[:, :, past:past + L], k_new, rtol=2 ** -7, atol=1e-2)
    vback = vcc.cpu().transpose(2, 3)
    assert torch.equal(vback[:, :, past:past + L], v_new)
    assert torch.equal(vback[:, :, :past], vc[:, :, :past]) and torch.equal(vback[:, :, past + L:], vc[:, :, past + L:])
    assert torch.equal(kcc[:, :, :past].cpu(), kc[:, :, :past]) and torch.equal(kcc[:, :, past + L:].cpu(), kc[:, :, past + L:])


@pytest.mark.parametrize("nkv", [8, 16])
@pytest.mark.parametrize("past,cap,dev_past", [(2531, 2688, True), (300, 1664, False)])
def test_fused_oproj_gqa_is_bit_identical_to_two_launches_or_declines(ops, nkv, past, cap, dev_past):
    """k_attn_decode128_o with 32 query heads over 8 / 16 kv heads: bit-identical to p3v_attention_decode + p3v_gemv (the residual
    row, the attention output, the appended K / V), ten launches in a row -- or P3V_ERR_UNSUPPORTED from the launcher, never a
    wrong result."""
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T, n_split = cap, cap // 128
    qkv = g((1, (nh + 2 * nkv) * hd), 145).cuda()
    kc0, vc0 = g((B, nkv, T, hd), 146).cuda(), g((B, nkv, hd, T), 147).cuda()
    wo = (g((H, nh * hd), 148) * 0.05).cuda()
    x0 = g((1, H), 149).cuda()
    gen = torch.Generator(device="cuda").manual_seed(150)
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda", generator=gen), torch.rand((B, 1, hd // 2), device="cuda", generator=gen)
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=True)
    hp = past - 40 if dev_past else past
    k1, v1, o1, x1 = kc0.clone(), vc0.clone(), torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode(qkv, cos, sin, 1, k1, v1, o1, B, L, nh, nkv, hd, hd ** -0.5, hp, T, ws, n_split, **kw)
    ops.gemv(o1.view(1, H), wo, ops.EPI_RESID_BF16, resid=x1, out=x1)
    # the two-launch result itself against fp32 attention over the repeated heads
    x = qkv.view(1, 1, nh + 2 * nkv, hd).transpose(1, 2).cpu()
    cs, sn = cos.cpu()[:, None], sin.cpu()[:, None]
    def rot(t):
        a, b = t[..., :hd // 2].float(), t[..., hd // 2:].float()
        return torch.cat([a * cs - b * sn, b * cs + a * sn], -1)
    q, k_new = rot(x[:, :nh]).to(BF16), rot(x[:, nh:nh + nkv]).to(BF16)
    kf = torch.cat([kc0.cpu()[:, :, :past], k_new], dim=2)
    vf = torch.cat([vc0.cpu().transpose(2, 3)[:, :, :past], x[:, nh + nkv:]], dim=2)
    w = (q.float() * hd ** -0.5) @ rep_kv(kf, nh).float().transpose(-1, -2)
    ref = (torch.softmax(w, -1) @ rep_kv(vf, nh).float()).transpose(1, 2).reshape(1, 1, nh * hd)
    close(o1, ref)
    for rep in range(10):
        k2, v2, x2 = kc0.clone(), vc0.clone(), x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        try:
            ops.attention_decode(qkv, cos, sin, 1, k2, v2, o2, B, L, nh, nkv, hd, hd ** -0.5, hp, T, ws, n_split, **kw,
                                 o_proj_w=wo, o_proj_x=x2, o_rearm=other)
        except RuntimeError as e:
            assert "UNSUPPORTED" in str(e).upper() or "-3" in str(e), e
            assert torch.equal(x2, x0) and torch.equal(k2, kc0)          # declined before anything was written
            return
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + gemv"
        assert torch.equal(k2, k1) and torch.equal(v2, v1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()


def _dequant(u8, sc, axis):
    return (u8.float() - 128.0) * sc.unsqueeze(axis)


@pytest.mark.parametrize("nh,nkv", GQA)
@pytest.mark.parametrize("past,L,n_split,fused", [(200, 1, 3, False), (200, 1, 4, False), (62, 5, 4, True), (200, 1, 2, True),
                                                   (126, 5, 2, True), (30, 16, 2, False)])
def test_kv_quantize_and_q8_decode_gqa(ops, orc, nh, nkv, past, L, n_split, fused):
    """int8 KV with fewer kv heads than query heads: the q8 decode attention equals the oracle attention over the DEQUANTISED
    nkv-head cache (multi-tile single-wave, single-tile 4-wave and k_attn_decode128_q8 plans), appended rows stored quantised."""
    B, hd, T = 2, 96, 256
    k, v = g((B, nkv, T, hd), 90), g((B, nkv, T, hd), 91)
    k8 = torch.full((B, nkv, T, hd), 128, dtype=torch.uint8).cuda()
    v8 = torch.full((B, nkv, hd, T), 128, dtype=torch.uint8).cuda()
    ksc, vsc = torch.ones((B, nkv, T)).cuda(), torch.ones((B, nkv, T)).cuda()
    ops.kv_quantize(k.cuda(), v.transpose(2, 3).contiguous().cuda(), k8, v8, ksc, vsc, 0, past)
    kd, vd = _dequant(k8.cpu(), ksc.cpu(), -1), _dequant(v8.cpu().transpose(2, 3), vsc.cpu(), -1)
    qkv = g((B * L, (nh + 2 * nkv) * hd), 92)
    cos, sin, cos_ref, sin_ref = _rope(ops, orc, B, T, hd)
    out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    for _ in range(2 if fused else 1):
        out.fill_(float("nan"))
        ops.attention_decode_q8(qkv.cuda(), cos[:, past:], sin[:, past:], T, k8, v8, ksc, vsc, out, B, L, nh, nkv, hd, hd ** -0.5, past,
                                T, ws, n_split, merge_in_launch=fused)
    assert (ws.view(torch.int32) == -1).all()
    x = qkv.view(B, L, nh + 2 * nkv, hd).transpose(1, 2)
    cs, sn = cos_ref[:, :, past:past + L], sin_ref[:, :, past:past + L]
    q = orc.rotate_half(x[:, :nh], cs, sn).to(BF16)
    k_new = orc.rotate_half(x[:, nh:nh + nkv], cs, sn).to(BF16)

    def qdq(t):
        sc = t.float().abs().amax(-1, keepdim=True) / 127
        return torch.round(t.float() / sc) * sc
    kf = torch.cat([kd[:, :, :past], qdq(k_new)], dim=2)
    vf = torch.cat([vd[:, :, :past], qdq(x[:, nh + nkv:])], dim=2)
    pad = torch.zeros(B, dtype=torch.int32)
    ref = attn_ref(orc, q, kf, vf, hd ** -0.5, allowed_mask(B, L, past, pad)).transpose(1, 2).reshape(B, L, nh * hd)
    close(out, ref)
    s_new = k_new.float().abs().amax(-1) / 127
    assert torch.allclose(ksc.cpu()[:, :, past:past + L], s_new, rtol=1e-6)
    assert (k8.cpu()[:, :, past:past + L].float() - (torch.round(k_new.float() / s_new[..., None]) + 128)).abs().max() <= 1
