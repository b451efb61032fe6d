"""Operator layer: torch tensors in, HIP kernels (via the C ABI) underneath.

torch is used for device memory and the current stream only; every function
here launches a hand-written gfx950 kernel from ``csrc/`` and nothing else.
Each op names the MLX call site of the reference it replaces.
"""
import contextlib
import ctypes as C

import torch

from . import _lib as L
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_QGELU, EPI_BIAS_RESID_F32, EPI_F32, EPI_NONE,  # noqa: F401
                   EPI_PATCH, EPI_RESID_BF16, EPI_SILU_MUL)

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (the hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def embed_gather(ids, table, out=None):
    """nn.Embedding (phi.py:577); ids int32 [n], table bf16 [V,H] -> bf16 [n,H]."""
    _chk(ids, I32, "ids"), _chk(table, BF16, "table")
    n, (V, H) = ids.numel(), table.shape
    out = torch.empty((n, H), dtype=BF16, device=table.device) if out is None else out
    L.check(L.lib().p3v_embed_gather(_p(ids), _p(table), _p(out), n, H, V, _stream()), "embed_gather")
    return out


def rmsnorm(x, w, eps, out=None):
    """nn.RMSNorm (phi.py:478-479,571)."""
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().p3v_rmsnorm(_p(x), _p(w), _p(out), rows, H, float(eps), _stream()), "rmsnorm")
    return out


def layernorm(x, w, b, eps, out_f32=False, out=None):
    """nn.LayerNorm (phi.py:165,167,212): fp32 rows -> bf16 (or fp32, may be in place)."""
    _chk(x, F32, "x"), _chk(w, BF16, "w"), _chk(b, BF16, "b")
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, dtype=F32 if out_f32 else BF16, device=x.device)
    L.check(L.lib().p3v_layernorm(_p(x), _p(w), _p(b), _p(out), int(out_f32), rows, H, float(eps), _stream()), "layernorm")
    return out


def set_tuning(name, value):
    """Pin a launch-policy knob of the library (include/p3v.h: p3v_set_tuning); returns the previous value."""
    old = C.c_int(0)
    L.check(L.lib().p3v_get_tuning(name.encode(), C.byref(old)), f"get_tuning({name})")
    L.check(L.lib().p3v_set_tuning(name.encode(), int(value)), f"set_tuning({name})")
    return old.value


_gemm_ws = {}     # (device index, stream handle) -> grow-only byte buffer for p3v_gemm's split-K partials


_ws_owner = None     # (holder dict, frozen?) while a hipGraph is being built: the GRAPH owns its split-K workspace


@contextlib.contextmanager
def owned_gemm_workspace(holder, frozen):
    """Route every split-K workspace request inside the block to `holder["buf"]` (a dict the caller keeps alive as long as the
    graph).  frozen=False (the warm-up run): the buffer is allocated / grown as needed.  frozen=True (the capture): no allocation
    may happen under hipStreamBeginCapture, and the pointer is baked into the graph -- a request the warm-up did not size
    raises instead.  (the per-(device, stream) buffer below was first allocated on the capture's side stream,
    i.e. under capture, and a later regrow would have freed what the graph still writes to.)"""
    global _ws_owner
    prev, _ws_owner = _ws_owner, (holder, frozen)
    try:
        yield holder
    finally:
        _ws_owner = prev


def _gemm_workspace(device, nbytes):
    """The library never allocates (include/p3v.h): the caller owns the split-K workspace.  One grow-only torch buffer per
    (device, stream): it is live only between the two launches of one p3v_gemm call, so launches on one stream share it."""
    if _ws_owner is not None:
        holder, frozen = _ws_owner
        buf = holder.get("buf")
        if buf is None or buf.numel() < nbytes:
            if frozen:
                raise RuntimeError(f"split-K workspace of {nbytes} bytes requested under graph capture (warm-up sized "
                                   f"{0 if buf is None else buf.numel()})")
            buf = holder["buf"] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return buf
    key = (device.index, _stream())
    buf = _gemm_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        # the old buffer may still be read by launches already queued on this stream: torch's caching allocator keeps
        # freed blocks stream-ordered, so dropping the reference is safe
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _gemm_ws[key] = buf
    return buf


def gemm(a, w, epilogue=EPI_NONE, bias=None, resid=None, out=None, n_out=None, pos=None, patches_per_img=0, ldo=None):
    """nn.Linear / patch conv: out[M,N] = a[M,K] @ w[N,K]^T (+ epilogue), bf16 MFMA."""
    _chk(a, BF16, "a"), _chk(w, BF16, "w")
    M, K = a.shape
    N = n_out if n_out is not None else (w.shape[0] // 2 if epilogue == EPI_SILU_MUL else w.shape[0])
    if w.shape[1] != K:
        raise ValueError(f"gemm: K mismatch {a.shape} x {w.shape}")
    f32_out = epilogue in (EPI_BIAS_RESID_F32, EPI_PATCH, EPI_F32)
    if out is None:
        out = torch.empty((M, N), dtype=F32 if f32_out else BF16, device=a.device)
    ws_bytes = L.lib().p3v_gemm_ws_bytes(M, N, K, epilogue)
    ws = _gemm_workspace(a.device, ws_bytes) if ws_bytes else None
    args = L.GemmArgs(_p(a), _p(w), _p(out), _p(bias), _p(resid), _p(pos), M, N, K, a.stride(0), w.stride(0),
                      ldo if ldo is not None else N, epilogue, patches_per_img, _p(ws), ws.numel() if ws is not None else 0)
    L.check(L.lib().p3v_gemm(C.byref(args), _stream()), "gemm")
    return out


def gemm_resid_norm(a, w, resid, norm_w, eps, normed, out=None):
    """out = resid + bf16(a @ w^T) and normed = RMSNorm(out) * norm_w from the projection's own launch sequence (the K-slice GEMM +
    ONE reduction launch that also normalises).  Returns False, nothing launched, where the library does not split the shape
    (P3V_ERR_UNSUPPORTED): the caller runs gemm(..., EPI_RESID_BF16) and rmsnorm."""
    _chk(a, BF16, "a"), _chk(w, BF16, "w"), _chk(resid, BF16, "resid"), _chk(norm_w, BF16, "norm_w"), _chk(normed, BF16, "normed")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or resid.shape != (M, N) or normed.shape != (M, N) or norm_w.numel() != N:
        raise ValueError(f"gemm_resid_norm: shapes {a.shape} x {w.shape}, resid {resid.shape}, normed {normed.shape}")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=a.device)
    ws_bytes = L.lib().p3v_gemm_ws_bytes(M, N, K, EPI_RESID_BF16)
    if not ws_bytes:
        return False
    ws = _gemm_workspace(a.device, ws_bytes)
    args = L.GemmArgs(_p(a), _p(w), _p(out), None, _p(resid), None, M, N, K, a.stride(0), w.stride(0), out.stride(0), EPI_RESID_BF16, 0,
                      _p(ws), ws.numel())
    rc = L.lib().p3v_gemm_resid_norm(C.byref(args), _p(norm_w), float(eps), _p(normed), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return False
    L.check(rc, "gemm_resid_norm")
    return True


GEMV_MAX_M = 16


def gemv(x, w, epilogue=EPI_NONE, resid=None, norm_w=None, norm_eps=0.0, out=None):
    """Skinny nn.Linear for decode (M<=16) with optional fused RMSNorm / epilogue."""
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    M, K = x.shape
    N = w.shape[0] // 2 if epilogue == EPI_SILU_MUL else w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F32 if epilogue == EPI_F32 else BF16, device=x.device)
    args = L.GemvArgs(_p(x), _p(w), _p(out), _p(resid), _p(norm_w), float(norm_eps), M, N, K, epilogue)
    L.check(L.lib().p3v_gemv(C.byref(args), _stream()), "gemv")
    return out


def resample_u8(src, out_len, axis, coeffs, bounds):
    """One pass of Pillow's 8-bit bilinear resample (phi.py:301) along `axis` (0 = rows, 1 = columns) of a uint8 image
    [H, W, C]; coeffs [out_len, ksize] int32, bounds [out_len, 2] int32 (processor.pil_bilinear_coeffs)."""
    _chk(src, torch.uint8, "src"), _chk(coeffs, I32, "coeffs"), _chk(bounds, I32, "bounds")
    H, W, Cn = src.shape
    if axis == 1:
        out = torch.empty((H, out_len, Cn), dtype=torch.uint8, device=src.device)
        outer, in_len, inner = H, W, Cn
    else:
        out = torch.empty((out_len, W, Cn), dtype=torch.uint8, device=src.device)
        outer, in_len, inner = 1, H, W * Cn
    L.check(L.lib().p3v_resample_u8(_p(src), _p(out), outer, in_len, out_len, inner, _p(coeffs), _p(bounds), coeffs.shape[1],
                                    _stream()), "resample_u8")
    return out


def hd_preprocess(resized, top, hp, portrait, lut, hw, hi, ww, wi, out):
    """Padded / transposed-back / normalised HD image -> crop slots + global view of `out` [n_slots, 3, 336, 336] f32."""
    _chk(resized, torch.uint8, "resized"), _chk(lut, torch.float64, "lut"), _chk(out, F32, "out")
    L.check(L.lib().p3v_hd_preprocess(_p(resized), resized.shape[0], resized.shape[1], int(top), int(hp), int(bool(portrait)), _p(lut),
                                      _p(hw), _p(hi), _p(ww), _p(wi), _p(out), out.shape[0], _stream()), "hd_preprocess")
    return out


def lora_down(x, lora_a, out=None):
    """t = x @ lora_a  (first half of LoRALinear.__call__, phi.py:131): x [M,K] bf16, lora_a [K,r] f32 -> [M,r] f32."""
    _chk(x, BF16, "x"), _chk(lora_a, F32, "lora_a")
    M, K = x.shape
    r = lora_a.shape[1]
    if out is None:
        out = torch.empty((M, r), dtype=F32, device=x.device)
    L.check(L.lib().p3v_lora_down(_p(x), _p(lora_a), _p(out), M, K, r, _stream()), "lora_down")
    return out


def lora_up(y, t, lora_b, scale, epilogue=EPI_NONE, resid=None, out=None):
    """out = epilogue((y + scale * (t @ lora_b)).astype(bf16))  (phi.py:131-133): y [M,N] bf16 (frozen projection),
    t [M,r] f32, lora_b [r,N] f32.  epilogue: EPI_NONE / EPI_RESID_BF16 / EPI_SILU_MUL (out [M, N/2])."""
    _chk(y, BF16, "y"), _chk(t, F32, "t"), _chk(lora_b, F32, "lora_b")
    M, N = y.shape
    r = lora_b.shape[0]
    if out is None:
        out = torch.empty((M, N // 2 if epilogue == EPI_SILU_MUL else N), dtype=BF16, device=y.device)
    L.check(L.lib().p3v_lora_up(_p(y), _p(t), _p(lora_b), float(scale), epilogue, _p(resid), _p(out), M, N, r, _stream()), "lora_up")
    return out


def gemv_q4(x, w4, sb, epilogue=EPI_NONE, resid=None, norm_w=None, norm_eps=0.0, out=None):
    """`gemv` on 4-bit group-64 weights (device layout of weights.q4_repack): x [M,K] bf16, w4 [N,K/8] i32, sb [N,K/64] i32.
    M = 1: the streaming GEMV (fused RMSNorm allowed); 2 <= M <= 8: k_gemv8_q4 (fused RMSNorm allowed); 9 <= M <= 16: k_gemm_rows_q4
    (no fused norm)."""
    _chk(x, BF16, "x"), _chk(w4, I32, "w4"), _chk(sb, I32, "sb")
    M, K = x.shape
    N = w4.shape[0] // 2 if epilogue == EPI_SILU_MUL else w4.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F32 if epilogue == EPI_F32 else BF16, device=x.device)
    args = L.GemvQ4Args(_p(x), _p(w4), _p(sb), _p(out), _p(resid), _p(norm_w), float(norm_eps), M, N, K, epilogue)
    L.check(L.lib().p3v_gemv_q4(C.byref(args), _stream()), "gemv_q4")
    return out


def dequant_q4(w4, sb, out=None):
    """4-bit group-64 weights -> bf16 [N, K] (scale * q + bias, rounded once)."""
    _chk(w4, I32, "w4"), _chk(sb, I32, "sb")
    N, K = w4.shape[0], w4.shape[1] * 8
    if out is None:
        out = torch.empty((N, K), dtype=BF16, device=w4.device)
    L.check(L.lib().p3v_dequant_q4(_p(w4), _p(sb), _p(out), N, K, _stream()), "dequant_q4")
    return out


def quantize_fp8_rows(w):
    """bf16 [N,K] -> (u8 e4m3fn bit patterns [N,K], fp32 scale [N]); w ~ fp8 * scale.  Load-time weight prep."""
    wf = w.float()
    scale = (wf.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
    w8 = (wf / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return w8, scale.contiguous()


def gemv_fp8(x, w8, w_scale, epilogue=EPI_NONE, resid=None, norm_w=None, norm_eps=0.0, out=None):
    """`gemv` on fp8 weights (half the streamed bytes)."""
    _chk(x, BF16, "x"), _chk(w8, torch.uint8, "w8"), _chk(w_scale, F32, "w_scale")
    M, K = x.shape
    N = w8.shape[0] // 2 if epilogue == EPI_SILU_MUL else w8.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=F32 if epilogue == EPI_F32 else BF16, device=x.device)
    args = L.GemvF8Args(_p(x), _p(w8), _p(w_scale), _p(out), _p(resid), _p(norm_w), float(norm_eps), M, N, K, epilogue)
    L.check(L.lib().p3v_gemv_fp8(C.byref(args), _stream()), "gemv_fp8")
    return out


def dequant_fp8(w8, w_scale, out=None):
    """fp8 rows -> bf16 (scratch for the prefill GEMM)."""
    N, K = w8.shape
    out = torch.empty((N, K), dtype=BF16, device=w8.device) if out is None else out
    L.check(L.lib().p3v_dequant_fp8(_p(w8), _p(w_scale), _p(out), N, K, _stream()), "dequant_fp8")
    return out


def quant_fp8_rows(x, norm_w=None, norm_eps=0.0, out=None, scale=None):
    """bf16 rows -> (e4m3 bytes [M, K], fp32 scale [M]), one scale per row = max|h| / 448; with `norm_w` h is the
    RMSNorm of x (the fused activation quantiser of the W8A8 prefill projections)."""
    _chk(x, BF16, "x")
    M, K = x.shape
    out = torch.empty((M, K), dtype=torch.uint8, device=x.device) if out is None else out
    scale = torch.empty((M,), dtype=F32, device=x.device) if scale is None else scale
    L.check(L.lib().p3v_quant_fp8_rows(_p(x), _p(norm_w), float(norm_eps), _p(out), _p(scale), M, K, _stream()), "quant_fp8_rows")
    return out, scale


def gemm_fp8(a8, a_scale, w8, w_scale, epilogue=EPI_NONE, resid=None, out=None):
    """W8A8 nn.Linear on the fp8 matrix cores: out[M,N] = epilogue(a_scale[m] w_scale[n] sum_k a8[m,k] w8[n,k]), bf16 out."""
    _chk(a8, torch.uint8, "a8"), _chk(w8, torch.uint8, "w8"), _chk(a_scale, F32, "a_scale"), _chk(w_scale, F32, "w_scale")
    M, K = a8.shape
    N = w8.shape[0] // 2 if epilogue == EPI_SILU_MUL else w8.shape[0]
    if w8.shape[1] != K:
        raise ValueError(f"gemm_fp8: K mismatch {a8.shape} x {w8.shape}")
    out = torch.empty((M, N), dtype=BF16, device=a8.device) if out is None else out
    args = L.GemmF8Args(_p(a8), _p(a_scale), _p(w8), _p(w_scale), _p(out), _p(resid), M, N, K, a8.stride(0), w8.stride(0), N, epilogue)
    L.check(L.lib().p3v_gemm_fp8(C.byref(args), _stream()), "gemm_fp8")
    return out


def gemm_fp8_ok(M, N_rows, K, epilogue):
    """Shapes p3v_gemm_fp8 takes (N_rows = rows of the weight matrix)."""
    n = N_rows // 2 if epilogue == EPI_SILU_MUL else N_rows
    return K % 128 == 0 and n % (128 if epilogue == EPI_SILU_MUL else 256) == 0 and epilogue in (EPI_NONE, EPI_RESID_BF16, EPI_SILU_MUL)


def linear(x, w, epilogue=EPI_NONE, resid=None, out=None):
    """Dispatch a projection to the weight-streaming GEMV (M<=8) or the MFMA GEMM."""
    if x.shape[0] <= GEMV_MAX_M and x.shape[1] % 512 == 0 or x.shape[0] <= 8 and epilogue in (EPI_NONE, EPI_RESID_BF16, EPI_SILU_MUL, EPI_F32):
        return gemv(x, w, epilogue, resid=resid, out=out)
    return gemm(x, w, epilogue, resid=resid, out=out)


def rope_table(pos, inv_freq, scale):
    """SuRoPE tables (phi.py:494-504): pos [n] f32, inv_freq [half] f32 -> cos, sin [n, half] f32."""
    _chk(pos, F32, "pos"), _chk(inv_freq, F32, "inv_freq")
    n, half = pos.numel(), inv_freq.numel()
    cos = torch.empty((n, half), dtype=F32, device=pos.device)
    sin = torch.empty_like(cos)
    L.check(L.lib().p3v_rope_table(_p(pos), _p(inv_freq), float(scale), _p(cos), _p(sin), n, half, _stream()), "rope_table")
    return cos, sin


def rope_kv_append(qkv, cos_t, sin_t, q_out, k_dst, v_dst, B, Lq, nh, nkv, hd, past, dst_t, dst_off_is_past, tab_t=0,
                   tab_div=1, d_past=None, q_scale=1.0):
    """split + _rotate_half + KVCache append (phi.py:443-452, 542-548).  q_scale: rotated queries are multiplied by it
    before their rounding to bf16 (`attention(..., q_prescaled=True)` then skips the per-score multiply)."""
    L.check(L.lib().p3v_rope_kv_append(_p(qkv), _p(cos_t), _p(sin_t), _p(q_out), _p(k_dst), _p(v_dst), B, Lq, nh, nkv, hd,
                                       int(past), _p(d_past), dst_t, int(dst_off_is_past), tab_t, tab_div, float(q_scale),
                                       _stream()),
            "rope_kv_append")


def gemm_qkv(a, w, cos_t, sin_t, q_out, k_dst, v_dst, B, Lq, nh, nkv, hd, past, dst_t, dst_off_is_past, tab_t=0, tab_div=1,
             q_scale=1.0, bias=None):
    """qkv projection + head split + _rotate_half + KVCache append in the GEMM's own launches (include/p3v.h: p3v_gemm_qkv) ==
    gemm(a, w[, bias]) followed by rope_kv_append(...), bit for bit.  Returns False -- nothing launched -- when the library does not
    take the shape (short prompts, unaligned append offsets, ...): the caller then runs the two calls."""
    _chk(a, BF16, "a"), _chk(w, BF16, "w")
    M, K = a.shape
    g = L.GemmArgs(_p(a), _p(w), 0, _p(bias), 0, 0, M, w.shape[0], K, a.stride(0), w.stride(0), w.shape[0],
                   EPI_BIAS if bias is not None else EPI_NONE, 0, 0, 0)
    sp = L.QkvSplit(_p(cos_t), _p(sin_t), _p(q_out), _p(k_dst), _p(v_dst), B, Lq, nh, nkv, hd, int(past), dst_t, int(dst_off_is_past),
                    tab_t, tab_div, float(q_scale))
    rc = L.lib().p3v_gemm_qkv(C.byref(g), C.byref(sp), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return False
    L.check(rc, "gemm_qkv")
    return True


Q_PRESCALE = 1.4426950408889634          # log2(e): q_scale = Q_PRESCALE * softmax scale for q_prescaled attention


def attention(q, out, B, Lq, nh, nkv, hd, scale, causal, k_new=None, v_new=None, new_t=0, past=0, k_past=None, v_past=None,
              past_t=0, past_div=1, pad_len=None, pad_div=1, d_past=None, ws=None, n_split=0, new_is_cache=False,
              q_prescaled=False):
    """softmax((q*scale) k^T + mask) v (phi.py:454-457 / phi.py:148), mask never materialised.
    K tensors are [B, nkv, t, hd]; V tensors are TRANSPOSED [B, nkv, hd, t] (t = past_t / new_t).
    q_prescaled: q already carries scale * log2(e) (rope_kv_append's q_scale = Q_PRESCALE * scale)."""
    args = L.AttnArgs(_p(q), _p(k_past), _p(v_past), _p(k_new), _p(v_new), _p(out), _p(pad_len), _p(d_past), _p(ws),
                      B, Lq, nh, nkv, hd, int(past), past_t, past_div, new_t, pad_div, int(causal), float(scale), n_split, int(new_is_cache),
                      int(bool(q_prescaled)))
    L.check(L.lib().p3v_attention(C.byref(args), _stream()), "attention")
    return out


def attention_decode_can_fuse_oproj(B, Lq, nh, hd, n_split, cache_t, o_n, merge_in_launch):
    """Does the fused attention + o_proj + residual launch take this shape on this device (include/p3v.h)?"""
    return bool(L.lib().p3v_attention_decode_can_fuse_oproj(B, Lq, nh, hd, n_split, cache_t, o_n, int(bool(merge_in_launch))))


def attention_decode_q8_can_fuse_oproj(B, Lq, nh, hd, n_split, cache_t, o_n, merge_in_launch):
    """Does the fused int8-KV attention + fp8 o_proj + residual launch take this shape on this device (include/p3v.h)?"""
    return bool(L.lib().p3v_attention_decode_q8_can_fuse_oproj(B, Lq, nh, hd, n_split, cache_t, o_n, int(bool(merge_in_launch))))


def attention_decode(qkv, cos_new, sin_new, rope_bstride, k_cache, v_cache, out, B, Lq, nh, nkv, hd, scale, past, cache_t, ws,
                     n_split, pad_len=None, d_past=None, merge_in_launch=False, o_proj_w=None, o_proj_x=None, o_rearm=None,
                     o_proj_sb=None):
    """Fused decode-step attention: head split + RoPE + KV append + split-KV attention + merge.
    cos_new/sin_new: rows of the new positions, row (b, r) at b*rope_bstride + r.
    merge_in_launch: the split partials are merged inside the attention launch (ws: see `attention_ws`).
    o_proj_w / o_proj_x / o_rearm: also the layer's o_proj + residual, x += bf16(W_o . out), in the same launch (`out` must be
    all 0xFF on entry; `o_rearm`, the other layer parity's output buffer, is set to 0xFF): see p3v_attn_decode_args_t.
    o_proj_sb: o_proj_w is then the 4-bit group-64 W4 [o_n, K / 8] (int32) of `gemv_q4` and o_proj_sb its scale | bias words."""
    if o_proj_w is not None:
        if o_proj_sb is not None:
            _chk(o_proj_w, I32, "o_proj_w"), _chk(o_proj_sb, I32, "o_proj_sb")
            if tuple(o_proj_w.shape) != (o_proj_sb.shape[0], nh * hd // 8) or o_proj_sb.shape[1] != nh * hd // 64:
                raise ValueError("fused o_proj (4-bit): o_proj_w is [o_n, K / 8], o_proj_sb [o_n, K / 64]")
        else:
            _chk(o_proj_w, BF16, "o_proj_w")
            if o_proj_w.dim() != 2 or o_proj_w.shape[1] != nh * hd:      # (the kernel reads o_n rows of n_heads * hd bf16)
                raise ValueError(f"fused o_proj: o_proj_w must be [o_n, {nh * hd}], got {tuple(o_proj_w.shape)}")
        _chk(o_proj_x, BF16, "o_proj_x"), _chk(o_rearm, BF16, "o_rearm")
        if out.numel() < nh * hd or o_rearm.numel() < nh * hd:
            raise ValueError("fused o_proj: out and o_rearm are [n_heads * hd] rows")
        if o_proj_x.numel() < o_proj_w.shape[0]:                         # x[0 : o_n] is updated in place
            raise ValueError(f"fused o_proj: o_proj_x holds {o_proj_x.numel()} values, the projection writes {o_proj_w.shape[0]}")
    args = L.AttnDecArgs(_p(qkv), _p(cos_new), _p(sin_new), _p(k_cache), _p(v_cache), _p(out), _p(pad_len), _p(d_past), _p(ws),
                         B, Lq, nh, nkv, hd, int(past), cache_t, rope_bstride, n_split, float(scale), int(bool(merge_in_launch)),
                         _p(o_proj_w), _p(o_proj_x), _p(o_rearm), 0 if o_proj_w is None else o_proj_w.shape[0], _p(o_proj_sb))
    L.check(L.lib().p3v_attention_decode(C.byref(args), _stream()), "attention_decode")
    return out


def kv_quantize(k, vt, k8, v8t, k_scale, v_scale, t0, n_tok):
    """bf16 K [B,nkv,Ts,hd] / V^T [B,nkv,hd,Ts] rows [t0,t0+n) -> offset-binary u8 caches + per-token scales."""
    B, nkv, src_t, hd = k.shape
    L.check(L.lib().p3v_kv_quantize(_p(k), _p(vt), _p(k8), _p(v8t), _p(k_scale), _p(v_scale), B * nkv, hd, src_t,
                                    k8.shape[2], int(t0), int(n_tok), _stream()), "kv_quantize")


def kv_quantize_mlx4(k, vt, k4, v4, k_sb, v_sb, n_tok, qkv=None, cos_t=None, sin_t=None, nh=0, past=0, tab_t=0, tab_div=1):
    """The reference's own prompt-cache format (phi.py:528-540: mx.quantize, group 32, 4 bits): tokens [0, n_tok) of bf16
    K [B,nkv,T,hd] / V^T [B,nkv,hd,T] -> codes [B,nkv,n_tok,hd/32,4] int32 + (scale, bias) [B,nkv,n_tok,hd/32,2] fp32, and the
    cache rows rewritten with the dequantised values.  qkv (+ cos_t / sin_t, nh): quantise the keys from their exact fp32 values,
    recomputed from the projection output (the reference's keys are fp32), instead of the bf16 cache rows."""
    _chk(k, BF16, "k"), _chk(vt, BF16, "vt"), _chk(k4, I32, "k4"), _chk(v4, I32, "v4"), _chk(k_sb, F32, "k_sb"), _chk(v_sb, F32, "v_sb")
    B, nkv, T, hd = k.shape
    if tuple(vt.shape) != (B, nkv, hd, T) or tuple(k4.shape) != (B, nkv, n_tok, hd // 32, 4) or tuple(k_sb.shape) != (B, nkv, n_tok, hd // 32, 2) \
            or k4.shape != v4.shape or k_sb.shape != v_sb.shape:
        raise ValueError("kv_quantize_mlx4: shapes")
    if qkv is not None:
        _chk(qkv, BF16, "qkv"), _chk(cos_t, F32, "cos_t"), _chk(sin_t, F32, "sin_t")
        if tuple(qkv.shape) != (B * n_tok, (nh + 2 * nkv) * hd):
            raise ValueError("kv_quantize_mlx4: qkv must be the projection output of exactly the n_tok tokens being quantised")
    L.check(L.lib().p3v_kv_quantize_mlx4(_p(k), _p(vt), _p(k4), _p(v4), _p(k_sb), _p(v_sb), B * nkv, hd, T, int(n_tok), _p(qkv), _p(cos_t),
                                         _p(sin_t), nh, nkv, int(past), tab_t, tab_div, _stream()), "kv_quantize_mlx4")


def kv_dequantize(k8, v8t, k_scale, v_scale, k, vt, n_tok):
    """offset-binary u8 caches -> bf16 K [B,nkv,Td,hd] / V^T [B,nkv,hd,Td], tokens [0, n_tok)."""
    B, nkv, src_t, hd = k8.shape
    L.check(L.lib().p3v_kv_dequantize(_p(k8), _p(v8t), _p(k_scale), _p(v_scale), _p(k), _p(vt), B * nkv, hd, src_t, k.shape[2],
                                      int(n_tok), _stream()), "kv_dequantize")


def attention_decode_q8(qkv, cos_new, sin_new, rope_bstride, k8, v8t, k_scale, v_scale, out, B, Lq, nh, nkv, hd, scale, past,
                        cache_t, ws, n_split, pad_len=None, d_past=None, merge_in_launch=False, o_proj_w8=None, o_proj_scale=None,
                        o_proj_x=None, o_rearm=None):
    """`attention_decode` on the int8 KV cache (merge_in_launch: as there).  o_proj_w8 / o_proj_scale / o_proj_x / o_rearm: also
    the layer's o_proj on e4m3 weights + residual in the same launch (as attention_decode's o_proj_*; see p3v.h)."""
    if o_proj_w8 is not None:
        _chk(o_proj_w8, torch.uint8, "o_proj_w8"), _chk(o_proj_scale, F32, "o_proj_scale"), _chk(o_proj_x, BF16, "o_proj_x"), _chk(o_rearm, BF16, "o_rearm")
        if tuple(o_proj_w8.shape) != (o_proj_scale.numel(), nh * hd) or out.numel() < nh * hd or o_rearm.numel() < nh * hd:
            raise ValueError("fused o_proj (fp8): o_proj_w8 is [o_n, n_heads * hd] with o_n row scales; out and o_rearm are rows")
        if o_proj_x.numel() < o_proj_w8.shape[0]:
            raise ValueError(f"fused o_proj (fp8): o_proj_x holds {o_proj_x.numel()} values, the projection writes {o_proj_w8.shape[0]}")
    args = L.AttnDecQ8Args(_p(qkv), _p(cos_new), _p(sin_new), _p(k8), _p(v8t), _p(k_scale), _p(v_scale), _p(out), _p(pad_len),
                           _p(d_past), _p(ws), B, Lq, nh, nkv, hd, int(past), cache_t, rope_bstride, n_split, float(scale), int(bool(merge_in_launch)),
                           _p(o_proj_w8), _p(o_proj_scale), _p(o_proj_x), _p(o_rearm), 0 if o_proj_w8 is None else o_proj_w8.shape[0])
    L.check(L.lib().p3v_attention_decode_q8(C.byref(args), _stream()), "attention_decode_q8")
    return out


def stage_rope(cos_t, sin_t, cos_out, sin_out, B, Lq, tab_t, past=0, d_past=None):
    """Copy the cos/sin rows of positions [past, past+L) into compact [B, L, half] buffers (graph-replayed decode)."""
    L.check(L.lib().p3v_stage_rope(_p(cos_t), _p(sin_t), int(past), _p(d_past), _p(cos_out), _p(sin_out), B, Lq, tab_t,
                                   cos_t.shape[-1], _stream()), "stage_rope")


def attention_ws_bytes(B, Lq, nh, hd, n_split):
    return int(L.lib().p3v_attention_ws_bytes(B, Lq, nh, hd, n_split))


def attention_ws(B, Lq, nh, hd, n_split, device):
    """The split-partial workspace in the state every decode-attention launch expects and leaves it in: all bytes 0xFF
    (the 'not written yet' sentinel of the in-launch merge, include/p3v.h)."""
    return torch.full((attention_ws_bytes(B, Lq, nh, hd, n_split) // 4,), -1, dtype=torch.int32, device=device).view(F32)


def im2col_patches(pix, patch, kpad, out=None):
    _chk(pix, F32, "pix")
    n, _, S, _ = pix.shape
    if out is None:
        out = torch.empty((n * (S // patch) ** 2, kpad), dtype=BF16, device=pix.device)
    L.check(L.lib().p3v_im2col_patches(_p(pix), _p(out), n, S, patch, kpad, _stream()), "im2col")
    return out


def clip_cls_rows(x, cls, pos):
    n, T, D = x.shape
    L.check(L.lib().p3v_clip_cls_rows(_p(x), _p(cls), _p(pos), n, T, D, _stream()), "clip_cls_rows")


def hd_merge(feats, sub_gn, glb_gn, h, w, grid, C_):
    """Phi3ImageEmbedding reshape/concat (phi.py:403-407) as one gather."""
    _chk(feats, F32, "feats")
    g2 = grid // 2
    n_out = h * g2 * (w * g2 + 1) + 1 + g2 * (g2 + 1)
    out = torch.empty((n_out, 4 * C_), dtype=BF16, device=feats.device)
    L.check(L.lib().p3v_hd_merge(_p(feats), _p(sub_gn), _p(glb_gn), _p(out), h, w, grid, C_, _stream()), "hd_merge")
    return out


def argmax(logits2d, out=None):
    """mx.argmax(logits[:, -1, :]) (phi_3_vision_mlx.py:386): bf16 rows, first max wins."""
    _chk(logits2d, BF16, "logits")
    rows, n = logits2d.shape
    out = torch.empty((rows,), dtype=I32, device=logits2d.device) if out is None else out
    L.check(L.lib().p3v_argmax(_p(logits2d), _p(out), rows, n, n, _stream()), "argmax")
    return out


def log_softmax(x):
    """nn.log_softmax over the last axis: x - bf16(logsumexp(x)), rounded to bf16 (the reference's composite)."""
    _chk(x, BF16, "x")
    n = x.shape[-1]
    y = torch.empty_like(x)
    L.check(L.lib().p3v_log_softmax(_p(x), _p(y), x.numel() // n, n, _stream()), "log_softmax")
    return y


def topk(x2d, k):
    """k largest per row ordered by (-value, index); replaces mx.argpartition (Q9)."""
    _chk(x2d, BF16, "x")
    rows, n = x2d.shape
    out = torch.empty((rows, k), dtype=I32, device=x2d.device)
    L.check(L.lib().p3v_topk(_p(x2d), _p(out), rows, n, k, n, _stream()), "topk")
    return out


def gemv_step_begin(tok, table, x_out, cos_t, sin_t, d_past, cos_out, sin_out, w, norm_w, norm_eps, out):
    """The step's FIRST projection with `step_begin` in its prologue (include/p3v.h: p3v_gemv_step): out = W . RMSNorm(table[tok]),
    x_out = table[tok], rotation rows of *d_past staged.  False (nothing launched) where the library keeps the separate launch."""
    _chk(table, BF16, "table"), _chk(x_out, BF16, "x_out"), _chk(tok, I32, "tok")
    B, tab_t, half = tok.numel(), cos_t.shape[-2], cos_t.shape[-1]
    st = L.GemvStep(_p(tok), _p(table), table.shape[0], _p(x_out), _p(cos_t), _p(sin_t), _p(cos_out), _p(sin_out), tab_t, half,
                    None, None, None, None, None, None, 0, _p(d_past))
    if isinstance(w, tuple) and w[0].dtype == I32:              # (4-bit weights, scale | bias words): p3v_gemv_q4_step
        w4, sb = w
        _chk(sb, I32, "sb")
        args = L.GemvQ4Args(None, _p(w4), _p(sb), _p(out), None, _p(norm_w), float(norm_eps), B, w4.shape[0], w4.shape[1] * 8, EPI_NONE)
        rc = L.lib().p3v_gemv_q4_step(C.byref(args), C.byref(st), _stream())
    elif isinstance(w, tuple):                                  # (e4m3 weights, fp32 row scales): p3v_gemv_fp8_step
        w8, ws = w
        _chk(w8, torch.uint8, "w8"), _chk(ws, F32, "w_scale")
        args = L.GemvF8Args(None, _p(w8), _p(ws), _p(out), None, _p(norm_w), float(norm_eps), B, w8.shape[0], w8.shape[1], EPI_NONE)
        rc = L.lib().p3v_gemv_fp8_step(C.byref(args), C.byref(st), _stream())
    else:
        _chk(w, BF16, "w")
        args = L.GemvArgs(None, _p(w), _p(out), None, _p(norm_w), float(norm_eps), B, w.shape[0], w.shape[1], EPI_NONE)
        rc = L.lib().p3v_gemv_step(C.byref(args), C.byref(st), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return False
    L.check(rc, "gemv_step(begin)")
    return True


def gemv_step_end(x, w, norm_w, norm_eps, out, next_tok, tok, history, d_step, d_past, ticket, amax_ws):
    """The step's LAST projection (final norm + lm_head) with `step_end` in its epilogue: logits -> out, their arg-max -> next_tok /
    tok / history[:, *d_step], counters bumped -- by the last workgroup to finish.  False where the library keeps the separate launch."""
    _chk(x, BF16, "x"), _chk(out, BF16, "out")
    if amax_ws.numel() * amax_ws.element_size() < L.GEMV_STEP_WS_BYTES:
        raise ValueError("gemv_step_end: amax_ws smaller than P3V_GEMV_STEP_WS_BYTES")
    M, K = x.shape
    st = L.GemvStep(None, None, 0, None, None, None, None, None, 0, 0,
                    _p(next_tok), _p(tok), _p(history), _p(d_step), _p(ticket), _p(amax_ws), history.shape[1], _p(d_past))
    if isinstance(w, tuple) and w[0].dtype == I32:              # (4-bit weights, scale | bias words)
        w4, sb = w
        _chk(sb, I32, "sb")
        args = L.GemvQ4Args(_p(x), _p(w4), _p(sb), _p(out), None, _p(norm_w), float(norm_eps), M, w4.shape[0], K, EPI_NONE)
        rc = L.lib().p3v_gemv_q4_step(C.byref(args), C.byref(st), _stream())
    elif isinstance(w, tuple):                                  # (e4m3 weights, fp32 row scales)
        w8, ws = w
        _chk(w8, torch.uint8, "w8"), _chk(ws, F32, "w_scale")
        args = L.GemvF8Args(_p(x), _p(w8), _p(ws), _p(out), None, _p(norm_w), float(norm_eps), M, w8.shape[0], K, EPI_NONE)
        rc = L.lib().p3v_gemv_fp8_step(C.byref(args), C.byref(st), _stream())
    else:
        _chk(w, BF16, "w")
        args = L.GemvArgs(_p(x), _p(w), _p(out), None, _p(norm_w), float(norm_eps), M, w.shape[0], K, EPI_NONE)
        rc = L.lib().p3v_gemv_step(C.byref(args), C.byref(st), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return False
    L.check(rc, "gemv_step(end)")
    return True


def add_i32(x, delta):
    L.check(L.lib().p3v_add_i32(_p(x), x.numel(), int(delta), _stream()), "add_i32")


def step_begin(tok, table, x_out, cos_t, sin_t, d_past, cos_out, sin_out, zero_buf=None):
    """Head of a replayed greedy step: embedding rows of `tok` + rotation rows of position *d_past (one launch);
    `zero_buf` (int32) is cleared as well (the in-launch producer flags of the step's fused launches)."""
    B, tab_t, half = tok.numel(), cos_t.shape[-2], cos_t.shape[-1]
    L.check(L.lib().p3v_step_begin(_p(tok), _p(table), _p(x_out), _p(cos_t), _p(sin_t), _p(d_past), _p(cos_out), _p(sin_out),
                                   B, table.shape[1], table.shape[0], tab_t, half, _p(zero_buf),
                                   0 if zero_buf is None else zero_buf.numel(), _stream()), "step_begin")


def step_end(logits, next_tok, tok, history, d_step, d_past, ticket):
    """Tail of a replayed greedy step: argmax + history/tok bookkeeping + counters (one launch)."""
    B, n = logits.shape[0], logits.shape[-1]
    L.check(L.lib().p3v_step_end(_p(logits), _p(next_tok), _p(tok), _p(history), _p(d_step), _p(d_past), _p(ticket), B, n,
                                 history.shape[1], _stream()), "step_end")


def store_token(tok, history, d_step, tok_next=None):
    B, max_steps = history.shape
    L.check(L.lib().p3v_store_token(_p(tok), _p(history), _p(d_step), _p(tok_next), B, max_steps, _stream()), "store_token")


class Graph:
    """hipGraph capture/replay of a launch sequence on the current stream."""

    def __init__(self):
        self.exec = C.c_void_p()

    def begin(self):
        L.check(L.lib().p3v_graph_begin(_stream()), "graph_begin")

    def end(self):
        L.check(L.lib().p3v_graph_end(_stream(), C.byref(self.exec)), "graph_end")

    def launch(self):
        L.check(L.lib().p3v_graph_launch(self.exec, _stream()), "graph_launch")

    def __del__(self):
        try:
            if self.exec:
                L.lib().p3v_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        L.check(L.lib().p3v_event_create(C.byref(self.h)), "event_create")

    def record(self):
        L.check(L.lib().p3v_event_record(self.h, _stream()), "event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float()
        L.check(L.lib().p3v_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "event_elapsed")
        return ms.value

    def __del__(self):
        try:
            L.lib().p3v_event_destroy(self.h)
        except Exception:
            pass


def device_props(device=0):
    p = L.Props()
    L.check(L.lib().p3v_device_props(device, C.byref(p)), "device_props")
    return {f: (getattr(p, f).decode() if f == "arch" else getattr(p, f)) for f, _ in L.Props._fields_}
