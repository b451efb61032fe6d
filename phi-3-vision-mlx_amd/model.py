"""Phi-3-Vision / Phi-3-mini-128K on MI355X: the model object behind
`generate()/choose()/constrain()`.

Drop-in for the reference's `Phi3VForCausalLM` / `Phi3ForCausalLM`
(reference phi.py:565-617): same call signature and return contract

    model(input_ids, pixel_values=None, image_sizes=None, positions=None, cache=None,
          pids=None, mask=None, max_tokens=0, advance_offset=None, n_beam=1) -> (logits, cache)

but every tensor op is a hand-written gfx950 kernel from ``csrc/`` (see ops.py).
This module only owns device buffers and sequences launches.

HBM layout
  weights      bf16 [out, in] row-major (HF layout, K contiguous = MFMA fragment order)
  KV cache     per layer K bf16 [B, n_kv, Tp, hd] and V TRANSPOSED bf16 [B, n_kv, hd, Tp]
               (Tp = prompt + max_tokens rounded up to 64, allocated once): both are then
               k-contiguous MFMA operands for QK^T and PV with plain 16-byte loads
  RoPE tables  fp32 cos/sin [B, T, hd/2] built once per prompt (one short/long choice, Q2)
  residual     decoder: bf16 [B*L, H]; ViT: fp32 [crops, 577, 1024]
Dtype flow differs from the reference only where stated in DESIGN.md
(bf16 K cache / bf16 MFMA operands instead of fp32 attention), within the
tolerance the parity tests state.
"""
import math
import copy
import os

import numpy as np
import torch

from . import ops
from .config import head_dim, is_vision, rope_scaling_factor
from .weights import Q4Weight, mlx_quantize, q4_repack
from .ops import (BF16, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_QGELU, EPI_BIAS_RESID_F32, EPI_NONE, EPI_PATCH,
                  EPI_RESID_BF16, EPI_SILU_MUL, F32, I32)

import functools
import weakref


def _on_device(fn):
    """Run a model entry point with the model's GPU as the current device: ops.py launches on
    `torch.cuda.current_stream()` of the CURRENT device, and a thread other than the one that loaded the model (the
    server's engine thread) starts on device 0."""
    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapped


V_PREFIX = "model.vision_embed_tokens.img_processor.vision_model."
E_PREFIX = "model.vision_embed_tokens."


class CacheState:
    """All layers' KV cache + per-prompt RoPE tables / pad lengths.

    Mirrors reference KVCache (phi.py:509-548) semantics: `offset` auto-advances
    by L, can be rewound (`advance_offset`), `max_tokens < 1` disables caching,
    `n_beam > 1` reads the cache without writing it."""

    def __init__(self, cfg, B, S, max_tokens, device):
        self.B, self.S, self.max_tokens = B, S, max_tokens
        self.T = S + max(max_tokens, 0)
        self.quantized = bool(getattr(cfg, "use_quantized_cache", False))
        # cache_format="mlx4" (opt-in): the reference's OWN quantised cache -- MLX 4-bit group-32 codes of the PROMPT's K / V, later
        # tokens unquantised (phi.py:528-540).  The codes are kept in the reference's format (k4 / v4 + scale | bias), and the
        # bf16 cache rows hold their dequantised values, which is what every later call of the reference attends on: the decode
        # path is then the plain bf16 one.  (Default for quantize_cache=True: int8 for all tokens, BASELINE config 5.)
        self.mlx4 = self.quantized and getattr(cfg, "cache_format", "int8") == "mlx4"
        if self.quantized and getattr(cfg, "cache_format", "int8") not in ("int8", "mlx4"):
            raise ValueError(f"cache_format must be 'int8' or 'mlx4', got {cfg.cache_format!r}")
        if self.mlx4:
            self.quantized = False
        # row/column stride of the caches: whole 128-key tiles (the decode attention takes one 128-key tile per workgroup
        # when the capacity allows it)
        gran = 128
        self.Tp = (self.T + gran - 1) // gran * gran
        if (self.Tp // gran) % 2 == 0:
            # V^T rows are Tp * 2 bytes apart: at multiples of 512 B the 96 rows of a tile crowd the same memory channels
            # (measured decode step at B = 1: +3.5 % at Tp = 2560, +2 % at 3584, +1.4 % at 4096, +5.7 % at 33024 = 258 tiles
            # against the neighbouring lengths) -- an ODD number of 128-key tiles keeps the stride off that grid
            self.Tp += gran
        nl, nkv, hd = cfg.num_hidden_layers, cfg.num_key_value_heads, head_dim(cfg)
        if self.quantized:
            # int8 KV (quantize_cache=True): offset-binary bytes + one fp32 scale per (layer, row, head, token);
            # one bf16 K / V^T scratch layer holds the exact prompt keys while a layer's prefill attends (phi.py:531-533)
            self.k8 = torch.full((nl, B, nkv, self.Tp, hd), 128, dtype=torch.uint8, device=device)
            self.v8 = torch.full((nl, B, nkv, hd, self.Tp), 128, dtype=torch.uint8, device=device)
            self.ks = torch.ones((nl, B, nkv, self.Tp), dtype=F32, device=device)
            self.vs = torch.ones((nl, B, nkv, self.Tp), dtype=F32, device=device)
            self.k_tmp = torch.empty((B, nkv, self.Tp, hd), dtype=BF16, device=device)
            self.v_tmp = torch.zeros((B, nkv, hd, self.Tp), dtype=BF16, device=device)
        else:
            self.k = torch.empty((nl, B, nkv, self.Tp, hd), dtype=BF16, device=device)    # K   [.., t, hd]
            self.v = torch.zeros((nl, B, nkv, hd, self.Tp), dtype=BF16, device=device)    # V^T [.., hd, t] (zero: tail keys stay finite)
        if self.mlx4:
            g = hd // 32
            self.k4 = torch.zeros((nl, B, nkv, max(S, 1), g, 4), dtype=I32, device=device)
            self.v4 = torch.zeros_like(self.k4)
            self.k_sb = torch.zeros((nl, B, nkv, max(S, 1), g, 2), dtype=F32, device=device)
            self.v_sb = torch.zeros_like(self.k_sb)
            self.mlx4_tokens = 0                                # prompt tokens held as codes (set by the first call)
        self.offset = 0
        self.cos = self.sin = self.pad_len = None
        self.graphs = {}
        self.epoch = None                                       # model.epoch the graphs were captured under
        self.shared = {"dirty": False}                          # shared by the per-request views of a captured-prefill entry (copy.copy)

    def mark_dirty(self):
        """An unrecovered failed step may have left NaN rows beyond the offset and half-armed buffers in the decode graph: a
        captured-prefill entry built on these buffers must not be reused (model._prefill_captured drops it)."""
        self.shared["dirty"] = True

    def scrub(self, lo, hi):
        """Forget cache positions [lo, hi) that a FAILED step wrote (api.greedy_loop's recovery): a poisoned step leaves NaN there,
        and the decode attention multiplies every V^T column of a tile -- live or not -- by its (zero) probability."""
        hi = min(hi, self.Tp)
        if hi <= lo:
            return
        if self.quantized:
            self.v8[..., lo:hi] = 128
            self.k8[:, :, :, lo:hi] = 128
            self.vs[..., lo:hi] = 1.0
            self.ks[..., lo:hi] = 1.0
        else:
            self.v[..., lo:hi] = 0
            self.k[:, :, :, lo:hi] = 0


class LayerCache:
    """`cache[i]` view with the reference's `.offset` attribute."""

    def __init__(self, state, i):
        self.state, self.i = state, i

    @property
    def offset(self):
        return self.state.offset

    @offset.setter
    def offset(self, v):
        self.state.offset = int(v)


class Phi3VModel:
    def __init__(self, cfg, weights, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("Phi3VModel needs a GPU: the hot path is HIP-only (no CPU fallback)")
        ops.L.lib()                                  # fail loudly if libp3v.so is missing
        self.cfg, self.device = cfg, torch.device(device)
        self.vision = is_vision(cfg)
        self.w = {k: v.to(self.device, BF16).contiguous() for k, v in weights.items() if not isinstance(v, Q4Weight)}
        # 4-bit group-64 projections (MLX nn.quantize checkpoints / quantized_int4): device layout of p3v_gemv_q4
        self.w4 = {k: tuple(t.to(self.device) for t in q4_repack(*v)) for k, v in weights.items() if isinstance(v, Q4Weight)}
        self.hd = head_dim(cfg)
        if self.hd != 96:
            raise ValueError(f"decoder head_dim must be 96 (got {self.hd})")
        self.hidden_hook = None                      # fn(layer, x [B*L, H], B, L) after every decoder layer (diagnostics)
        self.w8 = {}
        self._rope_tables = {}                                   # (B, T, factors, ..) -> (cos, sin): shared read-only (see _new_state)
        self._vit_graphs, self._vit_seen = {}, {}                # captured vision tower per crop count (clip_forward)
        self.adapters = {}                           # weight key -> (lora_a, lora_b, scale), see set_adapters
        self._lora_tmp = {}                          # decode-sized (M <= 16) adapter scratch: captured graphs point at it
        self._lora_flat = None                       # ONE grow-only scratch for prefill-sized calls (not graph-captured)
        # A long-lived owner that may share the GPU with other streams / processes (server.py) sets this: the decode step then
        # uses no launch that needs all of its workgroups resident at once (the fused attention + o_proj launch, the in-launch
        # split-KV merge on grids larger than the machine) -- ADVICE r04.  One-shot generate() on an exclusive GPU keeps them.
        self.serving = os.environ.get("P3V_SHARED_GPU", "0") == "1"
        self.epoch = 0                               # bumped whenever captured decode graphs become stale
        self._prefill_graphs = {}                    # (S, max_tokens) -> captured short-prompt prefill (see _prefill_captured)
        self._prefill_seen = {}
        self._states = weakref.WeakSet()             # every live CacheState (their graphs bake pointers into this model)
        # quantize_model=True (fp8): prompt-sized projections run W8A8 on the fp8 MFMA unless fp8_activations=False
        # (then: dequantise to a bf16 scratch + bf16 MFMA, weight-only accuracy at bf16 prefill speed)
        self.fp8_act = bool(getattr(cfg, "fp8_activations", True)) and os.environ.get("P3V_FP8_PREFILL", "mfma") != "dequant"
        if getattr(cfg, "quantized_fp8", False):
            self._quantize_decoder_fp8()
        if getattr(cfg, "quantized_int4", False):
            self._quantize_decoder_q4()
        if self.w4 and not hasattr(self, "_deq"):
            n_max = max(v[0].shape[0] * v[0].shape[1] * 8 for v in self.w4.values())
            self._deq = torch.empty(n_max, dtype=BF16, device=self.device)  # one dequantised matrix (prefill GEMM scratch)
        if self.vision:
            self._prep_vision()

    # ------------------------------------------------------------------ fp8 weights (quantize_model=True)
    def _quantize_decoder_fp8(self):
        """Decoder projections + lm_head -> e4m3 bytes + per-row scale (7.4 GB -> 3.7 GB streamed per token).
        Embedding, norms, the ViT and the projector stay bf16 (prefill-only / tiny)."""
        names = [k for k in self.w if k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith("_proj.weight"))]
        for k in names:
            self.w8[k] = ops.quantize_fp8_rows(self.w.pop(k))
        n_max = max(v[0].shape[0] * v[0].shape[1] for v in self.w8.values())
        self._deq = torch.empty(n_max, dtype=BF16, device=self.device)      # one dequantised matrix (prefill GEMM scratch)

    # ------------------------------------------------------------------ LoRA adapters (use_adapter=True)
    @_on_device
    def set_adapters(self, adapters):
        """Attach LoRA adapters: {"model.layers.<i>.<target>.weight": (lora_a [in,r] f32, lora_b [r,out] f32, scale)}
        (weights.resolve_adapter).  An adapted projection runs unfused: materialised RMSNorm -> frozen projection with a
        plain epilogue -> p3v_lora_down / p3v_lora_up, which carries the residual / SiLU*up epilogue (phi.py:129-133)."""
        for k, (a, b, _) in adapters.items():
            w = self.w.get(k)
            shape = w.shape if w is not None else (self.w8[k][0].shape if k in self.w8 else (self.w4[k][0].shape[0], self.w4[k][0].shape[1] * 8))
            if tuple(a.shape[:1]) != (shape[1],) or tuple(b.shape[1:]) != (shape[0],):
                raise ValueError(f"LoRA shapes {tuple(a.shape)} x {tuple(b.shape)} do not fit {k} {tuple(shape)}")
        self.adapters = {k: (a.to(self.device, F32).contiguous(), b.to(self.device, F32).contiguous(), float(s))
                         for k, (a, b, s) in adapters.items()}
        self.epoch += 1                              # captured decode graphs bake the kernel sequence and the scratch
        for st in list(self._states):                # pointers in: drop them on EVERY live cache before the scratch goes
            st.graphs.clear()
        self._lora_tmp, self._lora_flat = {}, None
        self._prefill_graphs.clear()

    def _quantize_decoder_q4(self):
        """Decoder projections + lm_head -> 4-bit group-64 (the reference's nn.quantize(model, 64, 4)): 7.4 GB -> 2.1 GB
        streamed per token.  Quantised on the device with the MLX algorithm (weights.mlx_quantize)."""
        names = [k for k in self.w if k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith("_proj.weight"))]
        for k in names:
            self.w4[k] = tuple(t.to(self.device) for t in q4_repack(*mlx_quantize(self.w.pop(k))))

    def _proj(self, x, key, epilogue=EPI_NONE, resid=None, norm_w=None, out=None, h=None):
        """One projection of the decoder (+ its LoRA adapter, if one is attached)."""
        ad = self.adapters.get(key)
        if ad is None:
            return self._proj_frozen(x, key, epilogue, resid, norm_w, out, h)
        a, b, scale = ad
        M, K, N, r = x.shape[0], x.shape[1], b.shape[1], a.shape[1]
        if M <= ops.GEMV_MAX_M:                                 # decode-sized: persistent buffers (graph replays read them)
            def buf(tag, shape, dtype):
                t_ = self._lora_tmp.get((tag,) + shape)
                if t_ is None:
                    t_ = self._lora_tmp[(tag,) + shape] = torch.empty(shape, dtype=dtype, device=self.device)
                return t_
            hbuf, y, t = buf("h", (M, K), BF16), buf("y", (M, N), BF16), buf("t", (M, r), F32)
        else:                                                   # prefill-sized: carved out of one grow-only allocation,
            nb = (M * K + M * N) * 2 + M * r * 4 + 512          # not one set per prompt length (a server would leak VRAM)
            if self._lora_flat is None or self._lora_flat.numel() < nb:
                self._lora_flat = torch.empty(nb, dtype=torch.uint8, device=self.device)
            f, o1 = self._lora_flat, (M * K * 2 + 255) // 256 * 256
            o2 = o1 + (M * N * 2 + 255) // 256 * 256
            hbuf = f[:M * K * 2].view(BF16).view(M, K)
            y = f[o1:o1 + M * N * 2].view(BF16).view(M, N)
            t = f[o2:o2 + M * r * 4].view(F32).view(M, r)
        if norm_w is not None:                                  # the adapter needs the normalised input itself
            x = ops.rmsnorm(x, norm_w, self.cfg.rms_norm_eps, out=hbuf)
        self._proj_frozen(x, key, EPI_NONE, None, None, y, None)
        ops.lora_down(x, a, out=t)
        return ops.lora_up(y, t, b, scale, epilogue, resid=resid, out=out)

    def _proj_frozen(self, x, key, epilogue=EPI_NONE, resid=None, norm_w=None, out=None, h=None):
        """Weight-streaming kernel for skinny x, MFMA GEMM otherwise; bf16 or fp8 weights."""
        eps = self.cfg.rms_norm_eps
        M, K = x.shape
        q, q4 = self.w8.get(key), self.w4.get(key)
        skinny = M <= 8 or (M <= ops.GEMV_MAX_M and K % 512 == 0)
        if skinny and M > 8 and q is None and q4 is None and K % 64 == 0 and self.w[key].shape[0] % 64 == 0:
            # 9 .. 16 rows on bf16 weights: the 64-row tiles of the weight-streaming GEMM (p3v_gemm_skinny.hip) beat the 16-row MFMA
            # GEMV by a quarter of the step (B = 16 at 512 keys: 3.9 -> 3.1 ms); up to 8 rows k_gemv_mfma8 stays ahead
            skinny = False
        if q4 is not None and M == 1 and K in (3072, 8192):
            return ops.gemv_q4(x, q4[0], q4[1], epilogue, resid=resid, norm_w=norm_w, norm_eps=eps, out=out)
        if (q4 is not None and 2 <= M <= 16 and K in (3072, 8192) and os.environ.get("P3V_Q4_ROWS", "1") != "0"
                and (q4[0].shape[0] // (2 if epilogue == EPI_SILU_MUL else 1)) % 16 == 0):
            # 2 .. 16 rows straight on the 4-bit weights (round 6: k_gemv8_q4 / k_gemm_rows_q4; the reference runs QuantizedLinear at
            # every batch size, phi_3_vision_mlx.py:296) instead of dequantising the whole matrix into the bf16 scratch per call
            if norm_w is not None and (M > 8 or os.environ.get("P3V_Q4_ROWS8_NORM", "1") == "0"):
                x, norm_w = ops.rmsnorm(x, norm_w, eps, out=h), None       # 9 .. 16 rows: one p3v_rmsnorm launch in front
            return ops.gemv_q4(x, q4[0], q4[1], epilogue, resid=resid, norm_w=norm_w, norm_eps=eps, out=out)
        if q is not None and skinny and K in (3072, 8192):
            return ops.gemv_fp8(x, q[0], q[1], epilogue, resid=resid, norm_w=norm_w, norm_eps=eps, out=out)
        if q is not None and not skinny and self.fp8_act and ops.gemm_fp8_ok(M, q[0].shape[0], K, epilogue):
            # prompt-sized input: W8A8 on the fp8 matrix cores, activations quantised per token row (fused with the norm)
            a8, sa = ops.quant_fp8_rows(x, norm_w, eps)
            return ops.gemm_fp8(a8, sa, q[0], q[1], epilogue, resid=resid, out=out)
        if q4 is not None:
            w = ops.dequant_q4(q4[0], q4[1], out=self._deq[:q4[0].numel() * 8].view(q4[0].shape[0], q4[0].shape[1] * 8))
        elif q is not None:
            w = ops.dequant_fp8(q[0], q[1], out=self._deq[:q[0].numel()].view(q[0].shape))
        else:
            w = self.w[key]
        if skinny:
            return ops.gemv(x, w, epilogue, resid=resid, norm_w=norm_w, norm_eps=eps, out=out)
        if norm_w is not None:
            x = ops.rmsnorm(x, norm_w, eps, out=h)
        return ops.gemm(x, w, epilogue, resid=resid, out=out)

    def _proj_resid_norm(self, o, key, x, norm_w, h):
        """x += bf16(o @ W^T) and h = RMSNorm(x) * norm_w from the projection's own launches (ops.gemm_resid_norm); False -- nothing
        done -- where that does not apply (adapters, quantised weights, shapes the library does not run as K slices)."""
        M = o.shape[0]
        if (key in self.adapters or key not in self.w or M <= 8 or h.shape[0] < M
                or os.environ.get("P3V_RESID_NORM_FUSE", "1") == "0"):
            return False
        return ops.gemm_resid_norm(o, self.w[key], x, norm_w, self.cfg.rms_norm_eps, h[:M], out=x)

    # ------------------------------------------------------------------ vision tower
    def _prep_vision(self):
        c = self.cfg.clip
        D, P = c["hidden_size"], c["patch_size"]
        if D // c["num_attention_heads"] != 64:
            raise ValueError("CLIP head_dim must be 64")
        kk = 3 * P * P
        self.kpad = (kk + 63) // 64 * 64
        wp = torch.zeros((D, self.kpad), dtype=BF16, device=self.device)
        wp[:, :kk] = self.w[V_PREFIX + "embeddings.patch_embedding.weight"].reshape(D, kk)
        self.w_patch = wp
        self.clip_qkv = []
        for j in range(c["num_hidden_layers"] - 1):          # the last layer never runs (phi.py:219)
            q = V_PREFIX + f"encoder.layers.{j}.self_attn."
            wq = torch.cat([self.w[q + f"{n}_proj.weight"] for n in "qkv"], dim=0).contiguous()
            bq = torch.cat([self.w[q + f"{n}_proj.bias"] for n in "qkv"], dim=0).contiguous()
            self.clip_qkv.append((wq, bq))

    VIT_GRAPH_ENTRIES = 2

    def _clip_bufs(self, n):
        """Every intermediate of the tower for n crops (one allocation set per call on the eager path; owned by the graph entry otherwise)."""
        c, dev = self.cfg.clip, self.device
        D, P, I_ = c["hidden_size"], c["patch_size"], c["intermediate_size"]
        G = c["image_size"] // P
        T, nh = G * G + 1, c["num_attention_heads"]
        Tp = (T + 63) // 64 * 64
        return dict(
            patches=torch.empty((n * G * G, self.kpad), dtype=BF16, device=dev), x=torch.empty((n, T, D), dtype=F32, device=dev),
            q=torch.empty((n, nh, T, 64), dtype=BF16, device=dev), k=torch.empty((n, nh, Tp, 64), dtype=BF16, device=dev),
            v=torch.zeros((n, nh, 64, Tp), dtype=BF16, device=dev),              # V^T, zero tail
            o=torch.empty((n * T, D), dtype=BF16, device=dev), h=torch.empty((n * T, D), dtype=BF16, device=dev),
            qkv=torch.empty((n * T, 3 * D), dtype=BF16, device=dev), f=torch.empty((n * T, I_), dtype=BF16, device=dev))

    @_on_device
    def clip_forward(self, pix):
        """ClipModel.__call__ (phi.py:216-221) on live crops; pix f32 [n,3,336,336] -> f32 [n,577,D]
        (row 0 = CLS, which the caller skips).
        Round 6: the tower's ~280 launches cost the host as long to enqueue (25 - 30 us each in Python + ctypes) as the GPU to run, so a
        crop count seen for the SECOND time is captured as one hipGraph over buffers the entry owns (as the short-prompt prefill,
        _prefill_captured) and later images of that geometry cost one copy + one graph launch; the returned features are the
        entry's buffer -- the caller consumes them on the same stream before the next image can overwrite them.  Same kernels in
        the same order: bit-identical.  P3V_VIT_GRAPH=0 keeps eager launches."""
        n = pix.shape[0]
        w = self.w
        key = (n, w[V_PREFIX + "embeddings.position_embedding.weight"].data_ptr(), w[V_PREFIX + "encoder.layers.0.mlp.fc1.weight"].data_ptr())
        if os.environ.get("P3V_VIT_GRAPH", "1") == "0" or self.hidden_hook is not None:
            return self._clip_body(pix, self._clip_bufs(n))
        e = self._vit_graphs.get(key)
        if e is None:
            if self._vit_seen.get(key, 0) < 1:                    # first sight of this crop count: eager, remember it
                if len(self._vit_seen) > 64:
                    self._vit_seen.clear()
                self._vit_seen[key] = 1
                return self._clip_body(pix, self._clip_bufs(n))
            e = self._build_vit_graph(pix)
            if len(self._vit_graphs) >= self.VIT_GRAPH_ENTRIES:
                self._vit_graphs.pop(next(iter(self._vit_graphs)))
            self._vit_graphs[key] = e
        e["pix"].copy_(pix, non_blocking=True)
        e["graph"].launch()
        return e["bufs"]["x"]

    def _build_vit_graph(self, pix):
        dev = self.device
        e = dict(pix=torch.empty_like(pix), bufs=self._clip_bufs(pix.shape[0]), ws={})
        e["pix"].copy_(pix)
        with ops.owned_gemm_workspace(e["ws"], frozen=False):
            self._clip_body(e["pix"], e["bufs"])                 # warm-up (sizes whatever workspace the graph owns)
        torch.cuda.synchronize()
        graph = ops.Graph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.owned_gemm_workspace(e["ws"], frozen=True):
            graph.begin()
            self._clip_body(e["pix"], e["bufs"])
            graph.end()
        torch.cuda.current_stream().wait_stream(side)
        e["graph"] = graph
        return e

    def _clip_body(self, pix, B_):
        c, w = self.cfg.clip, self.w
        n, D, P = pix.shape[0], c["hidden_size"], c["patch_size"]
        G = c["image_size"] // P
        T, nh, eps = G * G + 1, c["num_attention_heads"], c["layer_norm_eps"]
        e = V_PREFIX + "embeddings."
        patches = ops.im2col_patches(pix, P, self.kpad, out=B_["patches"])
        x = B_["x"]
        pos = w[e + "position_embedding.weight"]
        ops.gemm(patches, self.w_patch, EPI_PATCH, out=x, n_out=D, pos=pos, patches_per_img=G * G, ldo=D)
        ops.clip_cls_rows(x, w[e + "class_embedding"], pos)
        x2 = x.view(n * T, D)
        ops.layernorm(x2, w[V_PREFIX + "pre_layrnorm.weight"], w[V_PREFIX + "pre_layrnorm.bias"], 1e-5, out_f32=True, out=x2)
        Tp = (T + 63) // 64 * 64
        q, k, v, o, h, qkv, f = (B_[t] for t in ("q", "k", "v", "o", "h", "qkv", "f"))
        for j in range(c["num_hidden_layers"] - 1):
            lp = V_PREFIX + f"encoder.layers.{j}."
            ops.layernorm(x2, w[lp + "layer_norm1.weight"], w[lp + "layer_norm1.bias"], eps, out=h)
            wq, bq = self.clip_qkv[j]
            # head split; the queries leave it multiplied by scale * log2(e) like the decoder's (the attention's softmax is then the
            # bare exp2: k_attn_prefill_dma<64, PRE> 59 -> 49 us per layer at 17 crops, tools/clip_attn_probe.py).  Round 6: the split
            # rides in the projection's epilogue here too (ops.gemm_qkv: Q, K rows and V^T columns straight from the accumulators) --
            # a crop is 577 tokens, so the 8-token V^T runs start one token later in every crop; the epilogue now stores them at
            # whatever offset they fall on (one dwordx4 at even offsets, three stores at odd ones, element-wise across a crop end).
            pre = os.environ.get("P3V_VIT_PLAIN_Q") != "1"
            qs = 64 ** -0.5 * ops.Q_PRESCALE if pre else 1.0
            if not (os.environ.get("P3V_QKV_FUSE", "1") != "0"
                    and ops.gemm_qkv(h, wq, None, None, q, k, v, n, T, nh, nh, 64, 0, Tp, False, q_scale=qs, bias=bq)):
                ops.gemm(h, wq, EPI_BIAS, bias=bq, out=qkv)
                ops.rope_kv_append(qkv, None, None, q, k, v, n, T, nh, nh, 64, 0, Tp, False, q_scale=qs)
            ops.attention(q, o, n, T, nh, nh, 64, 64 ** -0.5, False, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
            ops.gemm(o, w[lp + "self_attn.out_proj.weight"], EPI_BIAS_RESID_F32, bias=w[lp + "self_attn.out_proj.bias"],
                     resid=x2, out=x2)
            ops.layernorm(x2, w[lp + "layer_norm2.weight"], w[lp + "layer_norm2.bias"], eps, out=h)
            ops.gemm(h, w[lp + "mlp.fc1.weight"], EPI_BIAS_QGELU, bias=w[lp + "mlp.fc1.bias"], out=f)
            ops.gemm(f, w[lp + "mlp.fc2.weight"], EPI_BIAS_RESID_F32, bias=w[lp + "mlp.fc2.bias"], resid=x2, out=x2)
        return x

    def vision_embed(self, x, pixel_values, image_sizes, positions, L):
        """Phi3ImageEmbedding.__call__ (phi.py:393-416): ViT on the live crops
        (dead zero-padded crop slots are skipped, Q5), HD merge, projector, and
        the projector's second GEMM writes straight into the text embeddings."""
        w = self.w
        if torch.is_tensor(pixel_values):                                  # processor(return_mx=True): already f32 on the GPU
            pv = pixel_values.to(F32)
        else:
            pv = torch.as_tensor(np.asarray(pixel_values), dtype=F32)      # f64 -> f32 as mx.array does
        sizes = (np.asarray(image_sizes) // 336).tolist()
        positions = np.asarray(positions).tolist()
        live = [h * ww + 1 for h, ww in sizes]
        if len(live) == 1 and pv.is_contiguous():                          # one image: its live crops are a view, nothing to gather
            pix = pv[0, :live[0]].to(self.device)
        else:
            pix = torch.cat([pv[i, :n] for i, n in enumerate(live)], dim=0).contiguous().to(self.device)
        feats = self.clip_forward(pix)
        C_ = self.cfg.img_processor["image_dim_out"]
        G = self.cfg.clip["image_size"] // self.cfg.clip["patch_size"]
        H = self.cfg.hidden_size
        idx, crop0 = 0, 0
        for i, (h, ww) in enumerate(sizes):
            merged = ops.hd_merge(feats[crop0:crop0 + live[i]], w[E_PREFIX + "sub_GN"], w[E_PREFIX + "glb_GN"], h, ww, G, C_)
            t = ops.gemm(merged, w[E_PREFIX + "img_projection.0.weight"], EPI_BIAS_GELU, bias=w[E_PREFIX + "img_projection.0.bias"])
            r, c0 = positions[idx]
            cnt = merged.shape[0]
            dst = x[r * L + c0: r * L + c0 + cnt]
            ops.gemm(t, w[E_PREFIX + "img_projection.2.weight"], EPI_BIAS, bias=w[E_PREFIX + "img_projection.2.bias"], out=dst)
            idx += cnt
            crop0 += live[i]
        return x

    # ------------------------------------------------------------------ per-prompt state
    ROPE_TABLE_ENTRIES = 8

    def _new_state(self, B, S, max_tokens, pids, mask, shared_tables=True):
        cfg = self.cfg
        st = CacheState(cfg, B, S, max_tokens, self.device)
        self._states.add(st)
        L_all = S + max_tokens                                  # reference sizes tables with max_tokens as given
        half = self.hd // 2
        su = cfg.rope_scaling["long_factor"] if L_all > cfg.original_max_position_embeddings else cfg.rope_scaling["short_factor"]
        # Prompts without explicit position ids (every unpadded request) of one geometry rotate by the same table: it is built once
        # and shared read-only between their states (a server's requests repeat a few geometries; building it costs three small host ->
        # device copies and a launch in front of every prefill).  Slot states write per-row tables: they get their own.
        tkey = (B, st.T, tuple(float(v) for v in su), float(cfg.rope_theta), rope_scaling_factor(cfg), self.hd) if pids is None and shared_tables else None
        if tkey is not None and tkey in self._rope_tables:
            st.cos, st.sin = self._rope_tables[tkey]
            if mask is not None:
                st.pad_len = torch.as_tensor((np.asarray(mask) == 0).sum(axis=1).astype(np.int32)).to(self.device)
            return st
        inv_freq = 1.0 / (torch.tensor(su, dtype=F32) * (torch.tensor(float(cfg.rope_theta), dtype=F32)
                                                        ** (torch.arange(0, self.hd, 2, dtype=F32) / self.hd)))
        T = st.T
        if pids is None:
            pos = torch.arange(T, dtype=F32)[None].expand(B, T)
        else:
            p = torch.as_tensor(np.asarray(pids)).to(F32)
            ext = p[:, -1:] + 1 + torch.arange(T - p.shape[1], dtype=F32)[None]
            pos = torch.cat([p, ext], dim=1)
        pos = pos.contiguous().to(self.device)
        cos, sin = ops.rope_table(pos.view(-1), inv_freq.to(self.device), rope_scaling_factor(cfg))
        st.cos, st.sin = cos.view(B, T, half), sin.view(B, T, half)
        if tkey is not None:
            if len(self._rope_tables) >= self.ROPE_TABLE_ENTRIES:
                self._rope_tables.pop(next(iter(self._rope_tables)))
            self._rope_tables[tkey] = (st.cos, st.sin)
        if mask is not None:
            m = np.asarray(mask)
            st.pad_len = torch.as_tensor((m == 0).sum(axis=1).astype(np.int32)).to(self.device)
        return st

    # ------------------------------------------------------------------ slot state of the continuous-batching engine
    @_on_device
    def new_slot_state(self, slots, window):
        """A cache of `slots` batch rows and `window` columns that is not tied to one prompt batch (engine.py): every row
        has its OWN left padding and position table, so a request can be prefilled into a free row while the other rows
        are mid-generation.  All rows start empty (pad_len = window: every key masked).  The window fixes the RoPE regime
        of every row (phi.py:492: one choice per call from prompt + max_tokens): window <= 4096 -> short factors, beyond
        -> long factors; the engine admits only requests that would pick the same factors alone.  With
        `quantize_cache=True` the rows keep the int8 KV cache (BASELINE config 5)."""
        st = self._new_state(slots, 0, window, None, None, shared_tables=False)
        st.pad_len = torch.full((slots,), window, dtype=I32, device=self.device)
        st.slots = True
        return st

    @_on_device
    def prefill_slot(self, st, row, inputs, return_logits=False):
        """Prefill one request (a B = 1 `processor(...)` result) -- or n requests as one batch (`collate_requests` of them:
        equal lengths, or shorter rows left-padded inside the group) -- into the batch rows row .. row+n-1 of a slot state so
        that every row's LAST prompt token sits in column st.offset - 1: row i gets left padding pad_i = st.offset - S_i and
        position ids 0..S_i-1 from column pad_i on.  Keys left of pad_i (stale rows of an earlier occupant, the group's own
        padding) are masked by pad_len.  Returns the first greedy tokens (int32 [n, 1] on the device; with return_logits also
        the last-position logits).  Each row computes what a B = 1 run of its request computes (pad invariance)."""
        cfg = self.cfg
        ids = np.asarray(inputs["input_ids"])
        ids = ids[None] if ids.ndim == 1 else ids
        n, S = ids.shape
        lens = np.full(n, S, dtype=np.int64)
        if "mask" in inputs:
            m = np.asarray(inputs["mask"]).reshape(n, S)
            lens = m.sum(1).astype(np.int64)
            if not all(int(m[i, S - lens[i]:].sum()) == lens[i] for i in range(n)) or lens.min() < 1:
                raise ValueError("prefill_slot takes LEFT-padded rows")
        win = st.offset - S                                      # first column the group's batch writes
        if win < 0:
            raise ValueError(f"prompt of {S} tokens does not fit left of column {st.offset}")
        pads = torch.as_tensor(st.offset - lens, dtype=torch.int32)
        half = self.hd // 2
        su = cfg.rope_scaling["long_factor" if st.T > cfg.original_max_position_embeddings else "short_factor"]   # as _new_state
        inv_freq = 1.0 / (torch.tensor(su, dtype=F32)
                          * (torch.tensor(float(cfg.rope_theta), dtype=F32) ** (torch.arange(0, self.hd, 2, dtype=F32) / self.hd)))
        rows = slice(row, row + n)
        uniq = np.unique(lens)
        for L_ in uniq:                                          # one table per distinct length
            pos = (torch.arange(st.T, dtype=F32) - float(st.offset - L_)).clamp_min(0).to(self.device)
            cos, sin = ops.rope_table(pos, inv_freq.to(self.device), rope_scaling_factor(cfg))
            for i in np.nonzero(lens == L_)[0]:
                st.cos[row + i].copy_(cos.view(st.T, half)), st.sin[row + i].copy_(sin.view(st.T, half))
        st.pad_len[rows].copy_(pads.to(self.device))
        view = CacheState.__new__(CacheState)                   # these rows as an n-row cache at offset `win`
        view.__dict__.update(B=n, S=S, max_tokens=st.max_tokens, T=st.T, Tp=st.Tp, quantized=st.quantized, offset=win, graphs={},
                             epoch=self.epoch, cos=st.cos[rows], sin=st.sin[rows], pad_len=st.pad_len[rows],
                             fresh_rows=True)    # columns left of `win` hold nothing these rows may see (pad_len >= win)
        if st.quantized:
            view.__dict__.update(k8=st.k8[:, rows], v8=st.v8[:, rows], ks=st.ks[:, rows], vs=st.vs[:, rows],
                                 k_tmp=st.k_tmp[rows], v_tmp=st.v_tmp[rows])
        else:
            view.__dict__.update(k=st.k[:, rows], v=st.v[:, rows])
        kw = {k: v for k, v in inputs.items() if k in ("pixel_values", "image_sizes", "positions")}
        logits, _ = self(input_ids=ids, cache=[LayerCache(view, i) for i in range(cfg.num_hidden_layers)], full_logits=False, **kw)
        assert view.offset == st.offset
        tok = ops.argmax(logits[:, -1, :].contiguous())[:, None]
        return (tok, logits) if return_logits else tok

    @_on_device
    def decode_graph(self, st):
        """The captured greedy step of a state (built on first use): its `tok` / `next_tok` / `d_past` device buffers."""
        if st.epoch != self.epoch:
            st.graphs.clear()
            st.epoch = self.epoch
        g = st.graphs.get("greedy")
        if g is None:
            g = st.graphs["greedy"] = self._build_decode_graph(st)
            g["host_tok"] = None
        return g

    # ------------------------------------------------------------------ decoder stack
    def _alloc_bufs(self, B, L):
        cfg = self.cfg
        nh, nkv, hd, H, I = cfg.num_attention_heads, cfg.num_key_value_heads, self.hd, cfg.hidden_size, cfg.intermediate_size
        M, dev = B * L, self.device
        bufs = dict(
            q=torch.empty((B, nh, L, hd), dtype=BF16, device=dev), o=torch.empty((M, nh * hd), dtype=BF16, device=dev),
            qkv=torch.empty((M, (nh + 2 * nkv) * hd), dtype=BF16, device=dev), a=torch.empty((M, I), dtype=BF16, device=dev),
            h=torch.empty((M, H), dtype=BF16, device=dev), n_split=0, ws=None)
        return bufs

    def _plan_fused_oproj(self, bufs, B, L, T, quantized=False):
        """B = L = 1 decode: attention + o_proj + residual as ONE launch per layer where the library takes the shape -- on bf16 or
        MLX 4-bit weights and a bf16 cache (k_attn_decode128_o / _o4), or on e4m3 weights and the int8 cache (config 5:
        k_attn_decode128_q8<true>); p3v_attention.hip.  Long contexts (round 6), whose partials are merged by a launch of their own:
        that launch carries the o_proj (k_attn_combine_o).  The attention output then lives in two buffers that alternate layers use, both all-ones (= "not written
        yet") between launches; P3V_ATTN_FUSE_OPROJ=0 switches it off."""
        cfg = self.cfg
        can = ops.attention_decode_q8_can_fuse_oproj if quantized else ops.attention_decode_can_fuse_oproj
        o_key = "model.layers.0.self_attn.o_proj.weight"
        # The launch of layer i re-arms the OTHER buffer, so the two alternate cleanly only over an EVEN number of layers (an odd
        # stack would hand layer 0 of the next step the buffer the last layer just filled: stale words read as "written").
        # It also needs the GPU to itself (every workgroup resident at once): not for a server-owned model.
        # P3V_PROFILING=1: no launch of the step may wait for another workgroup of its own grid (whole-graph rocprofv3 --pmc passes
        # serialise / mask what they profile): separate o_proj launch here, separate merge launch in _split_plan
        ok = (os.environ.get("P3V_ATTN_FUSE_OPROJ", "1") != "0" and os.environ.get("P3V_PROFILING") != "1"
              and B == 1 and L == 1
              and cfg.num_hidden_layers % 2 == 0 and not self.serving
              and not self.adapters and (o_key in self.w8) == bool(quantized) and not (quantized and o_key in self.w4)
              # (without the in-launch merge -- long contexts: the MERGE launch carries the o_proj, k_attn_combine_o, round 6)
              and can(B, L, cfg.num_attention_heads, self.hd, bufs["n_split"], T, cfg.hidden_size, bool(bufs.get("attn_merge", False))))
        if not ok and bufs.get("attn_merge", False) and B == 1 and L == 1 and not quantized:
            # Short contexts (64-key one-tile plans: the in-launch form above is for 128-key tiles): attention that only writes its partials
            # + the merge launch carrying the o_proj beats attention with the in-launch merge + an o_proj launch (1.587 -> 1.570 ms per
            # step at 128 keys, 1.662 -> 1.652 at 1000; with the int8 cache it loses: 1.165 -> 1.179 at 128 keys, so not there)
            bufs["attn_merge"] = False
            self._plan_fused_oproj(bufs, B, L, T, quantized)
            if not bufs["fuse_o"]:
                bufs["attn_merge"] = True
            return
        bufs["fuse_o"] = bool(ok)
        if ok:
            for k in ("o_f", "o_f2"):                            # all-ones = "not written yet"
                bufs[k] = torch.full((1, cfg.num_attention_heads * self.hd), -1, dtype=torch.int16, device=self.device).view(BF16)

    def _split_plan(self, bufs, B, L, T, quantized=False, serving=False):
        """Split-KV plan for the decode-shaped attention (L <= 16): enough blocks to fill 256 CUs.
        serving: the plan is for a slot state of the continuous-batching engine (a long-lived server) -- the in-launch merge is
        then used only when EVERY workgroup of the launch is resident at once (no assumption about dispatch order at all)."""
        nh, hd = self.cfg.num_attention_heads, self.hd
        if L <= ops.L.DECODE_MAX_L:
            # 64-key tiles.  Up to ~4096 workgroups: ONE tile per 4-wave workgroup (the kernel then lasts a single
            # tile's dependency chain); beyond that ~768 single-wave workgroups (one resident round, each with its
            # next 24 KB tile in flight) walking several tiles
            tiles = -(-T // 64)
            tiles128 = T // 128 if T % 128 == 0 else 0           # T is the cache CAPACITY when the plan is for a captured graph
            if tiles128 and tiles > 16 and tiles128 <= 48 and B * nh * tiles128 <= 2048 \
                    and os.environ.get("P3V_ATTN_TILE128", "1") != "0":
                # one 128-key tile per workgroup: every workgroup of the launch resident at once (3 per CU), half the
                # partials to merge (k_attn_decode128).  Short caches keep 64-key tiles (more workgroups than CUs matters more)
                n_split = tiles128
            elif tiles <= 128 and B * nh * tiles <= 4096:
                n_split = tiles
            else:
                # (int8 KV: a tile is half the bytes and the single-wave kernel's 27.5 KB of LDS lets 5 workgroups share a CU,
                #  so ~1280 workgroups keep as many bytes in flight: 3.68 -> 3.40 ms/step at 32k)
                n_split = max(1, min(128, tiles, -(-(1280 if quantized else 768) // max(1, B * nh))))
            if os.environ.get("P3V_ATTN_NSPLIT"):
                n_split = int(os.environ["P3V_ATTN_NSPLIT"])
            bufs["n_split"] = n_split
            bufs["ws"] = ops.attention_ws(B, L, nh, hd, n_split, self.device)
            # in-launch split-KV merge (`merge_in_launch`): with the one-tile-per-workgroup plans, and with the
            # multi-tile streaming kernel when a head has at most 4 splits (B = 8: 3 splits -- the merge launch costs 4.9 us a
            # layer, the in-launch merge 0.5; with 24 splits at B = 1 / 32k the one merging workgroup per head is the slower
            # way: +1.6 % per step, +4.5 % at 8k)
            mode = os.environ.get("P3V_ATTN_FUSED_MERGE", "1")
            fused = (n_split in (tiles, tiles128) or n_split <= 4 or mode == "2") and n_split <= 48 and mode != "0"
            # The merging workgroup of a (row, head) waits -- bounded, NaN-poisoned on expiry -- for partials of workgroups that
            # the hardware dispatches before it (linear order; observed, not documented).  When the whole grid is resident at once
            # (128-key tiles: 3 workgroups per CU, 64-key: 5) nothing depends on that order.  A server takes the separate merge
            # launch wherever the grid is larger (B = 8: +4.9 us per layer); one-shot generate() keeps the faster in-launch form,
            # whose failure mode is loud (api._rows raises).  P3V_ATTN_FUSED_MERGE=2 forces it everywhere.
            if serving and fused and mode != "2":
                per_cu = 3 if n_split == tiles128 else 5
                fused = B * nh * n_split <= per_cu * ops.device_props(torch.device(self.device).index or 0)["cu_count"]
            if os.environ.get("P3V_PROFILING") == "1":           # (see _plan_fused_oproj)
                fused = False
            bufs["attn_merge"] = bool(fused)

    def _layers(self, x, st, B, L, past, n_beam, bufs=None, d_past=None, last_only=False, step_begin=None):
        """Phi3DecoderLayer stack (phi.py:473-485).  `d_past` (device int32) makes every
        position-dependent kernel read the cache length from HBM -> graph-replayable.
        last_only: the caller reads the LAST position only (prefill of generate / choose, Q8): the final layer still builds
        K / V for every position, but its o_proj and MLP run on the B last rows alone (0.4 ms of a 30 ms prefill);
        returns [B, H] then."""
        cfg, w = self.cfg, self.w
        nh, nkv, hd, eps = cfg.num_attention_heads, cfg.num_key_value_heads, self.hd, cfg.rms_norm_eps
        M = B * L
        skinny = M <= 8 or (M <= ops.GEMV_MAX_M and cfg.hidden_size % 512 == 0)   # weight-streaming projections
        scale = hd ** -0.5
        if bufs is None:
            bufs = self._alloc_bufs(B, L)
            self._split_plan(bufs, B, L, st.Tp, st.quantized, serving=getattr(st, "serving", False) or self.serving)   # the CAPACITY, as the captured graph plans: same kernel, same
                                                                # split boundaries -> eager and replayed steps agree bit for bit
            if L <= ops.L.DECODE_MAX_L and n_beam == 1:
                self._plan_fused_oproj(bufs, B, L, st.Tp, st.quantized)
        q, o, qkv, a, h, n_split, ws = (bufs[k] for k in ("q", "o", "qkv", "a", "h", "n_split", "ws"))
        mlx4 = getattr(st, "mlx4", False)
        if (st.quantized or mlx4) and n_beam > 1:
            raise NotImplementedError("Beam Search is not yet compatible with Quantized Cache")       # as phi.py:525
        mlx4_first = mlx4 and past == 0 and st.mlx4_tokens == 0 and L <= st.k4.shape[3]             # the call that fills the cache (phi.py:531-533)
        if n_beam > 1:                                          # beams: K/V of this call go to a scratch, cache is read-only
            Lp = (L + 7) // 8 * 8
            k_new = torch.empty((B, nkv, Lp, hd), dtype=BF16, device=self.device)
            v_new = torch.zeros((B, nkv, hd, Lp), dtype=BF16, device=self.device)
        normed_in = False                                       # `h` already holds the layer's normalised input
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            # Prompt-sized calls on bf16 weights: the qkv projection writes rotated Q, the K cache rows and the V^T cache columns from
            # its own epilogue (ops.gemm_qkv: bit-identical to the projection + rope_kv_append, one launch sequence instead of two).
            fused_qkv = False
            k_w = p + "self_attn.qkv_proj.weight"
            if (L > ops.L.DECODE_MAX_L and n_beam == 1 and k_w in w and k_w not in self.adapters and not mlx4_first
                    and (M >= 1024 or M <= 256) and os.environ.get("P3V_QKV_FUSE", "1") != "0"):
                kd, vd = (st.k_tmp, st.v_tmp) if st.quantized else (st.k[i], st.v[i])
                if not (st.quantized and past > 0 and not getattr(st, "fresh_rows", False)):
                    hn = h if normed_in else ops.rmsnorm(x, w[p + "input_layernorm.weight"], eps, out=h)
                    normed_in = True                            # (`h` holds the normalised input now, whatever the fused call answers)
                    fused_qkv = ops.gemm_qkv(hn, w[k_w], st.cos, st.sin, q, kd, vd, B, L, nh, nkv, hd, past, st.Tp, True, st.T, 1,
                                             q_scale=scale * ops.Q_PRESCALE)
            if i == 0 and step_begin is not None:               # replayed greedy step: the embedding gather + rotation-row staging
                sb = step_begin                                 # ride in this projection's prologue when the library takes the shape
                w_fold = w.get(k_w) if k_w in w else (self.w8.get(k_w) or self.w4.get(k_w))   # bf16, (e4m3, row scales) or (4-bit, scale | bias)
                if not (w_fold is not None and k_w not in self.adapters and os.environ.get("P3V_STEP_FOLD", "1") != "0"
                        and ops.gemv_step_begin(sb["tok"], sb["table"], x, st.cos, st.sin, d_past, sb["cos_o"], sb["sin_o"],
                                                w_fold, w[p + "input_layernorm.weight"], eps, qkv)):
                    ops.step_begin(sb["tok"], sb["table"], x, st.cos, st.sin, d_past, sb["cos_o"], sb["sin_o"])
                    self._proj(x, k_w, norm_w=w[p + "input_layernorm.weight"], out=qkv, h=h)
            elif not fused_qkv:
                if normed_in:                                   # `h` holds RMSNorm(x) already (the previous down_proj's reduction launch)
                    self._proj(h, p + "self_attn.qkv_proj.weight", out=qkv)
                else:
                    self._proj(x, p + "self_attn.qkv_proj.weight", norm_w=w[p + "input_layernorm.weight"], out=qkv, h=h)
            if st.quantized:
                if L <= ops.L.DECODE_MAX_L:
                    if d_past is not None:
                        rc, rs, rb = bufs["rope_cos"], bufs["rope_sin"], L
                    else:
                        rc, rs, rb = st.cos[:, past:], st.sin[:, past:], st.T
                    kq, o_i = {}, o
                    if bufs.get("fuse_o", False):               # + o_proj (e4m3) + residual in the same launch
                        o_i, o_other = (bufs["o_f"], bufs["o_f2"]) if i % 2 == 0 else (bufs["o_f2"], bufs["o_f"])
                        w8o = self.w8[p + "self_attn.o_proj.weight"]
                        kq = dict(o_proj_w8=w8o[0], o_proj_scale=w8o[1], o_proj_x=x, o_rearm=o_other)
                    ops.attention_decode_q8(qkv, rc, rs, rb, st.k8[i], st.v8[i], st.ks[i], st.vs[i], o_i, B, L, nh, nkv, hd, scale,
                                            past, st.Tp, ws, n_split, pad_len=st.pad_len, d_past=d_past,
                                            merge_in_launch=bufs.get("attn_merge", False), **kq)
                else:                                           # prefill: exact attention, quantised copy stored
                    if past > 0 and not getattr(st, "fresh_rows", False):   # long cached call (constrain with > 16 tokens): attend on a
                        ops.kv_dequantize(st.k8[i], st.v8[i], st.ks[i], st.vs[i], st.k_tmp, st.v_tmp, past)   # dequantised copy
                    if not fused_qkv:
                        ops.rope_kv_append(qkv, st.cos, st.sin, q, st.k_tmp, st.v_tmp, B, L, nh, nkv, hd, past, st.Tp, True, st.T, 1,
                                           q_scale=scale * ops.Q_PRESCALE)
                    ops.attention(q, o, B, L, nh, nkv, hd, scale, True, past=past, k_past=st.k_tmp, v_past=st.v_tmp,
                                  past_t=st.Tp, pad_len=st.pad_len, new_is_cache=True, q_prescaled=True)
                    ops.kv_quantize(st.k_tmp, st.v_tmp, st.k8[i], st.v8[i], st.ks[i], st.vs[i], past, L)
            elif n_beam > 1:
                ops.rope_kv_append(qkv, st.cos, st.sin, q, k_new, v_new, B, L, nh, nkv, hd, past, Lp, False, st.T, n_beam)
                ops.attention(q, o, B, L, nh, nkv, hd, scale, True, k_new=k_new, v_new=v_new, new_t=Lp, past=past,
                              k_past=st.k[i], v_past=st.v[i], past_t=st.Tp, past_div=n_beam, pad_len=st.pad_len,
                              pad_div=n_beam, ws=ws, n_split=n_split)
            elif L <= ops.L.DECODE_MAX_L:                       # decode-shaped step: one fused launch (+ merge)
                if d_past is not None:                          # graph replay: rows staged once per step by the caller
                    rc, rs, rb = bufs["rope_cos"], bufs["rope_sin"], L
                else:                                           # eager: views into the prompt tables at `past`
                    rc, rs, rb = st.cos[:, past:], st.sin[:, past:], st.T
                # (captured step: `past` is read from d_past; the host value passed along is a LOWER BOUND of it -- the kernel fetches
                #  tiles below it at once and lets the others wait for the length, so tiles beyond the live keys cost nothing)
                fuse_o = bufs.get("fuse_o", False)
                if fuse_o:                                      # + o_proj + residual in the same launch: x += bf16(W_o . o)
                    o_i, o_other = (bufs["o_f"], bufs["o_f2"]) if i % 2 == 0 else (bufs["o_f2"], bufs["o_f"])
                    q4o = self.w4.get(p + "self_attn.o_proj.weight")
                    kw_o = dict(o_proj_w=q4o[0], o_proj_sb=q4o[1]) if q4o is not None else dict(o_proj_w=w[p + "self_attn.o_proj.weight"])
                    ops.attention_decode(qkv, rc, rs, rb, st.k[i], st.v[i], o_i, B, L, nh, nkv, hd, scale,
                                         past if d_past is None else bufs.get("past_lb", -1), st.Tp, ws, n_split,
                                         pad_len=st.pad_len, d_past=d_past, merge_in_launch=bufs.get("attn_merge", False), o_proj_x=x,
                                         o_rearm=o_other, **kw_o)
                else:
                    ops.attention_decode(qkv, rc, rs, rb, st.k[i], st.v[i], o, B, L, nh, nkv, hd, scale,
                                         past if d_past is None else bufs.get("past_lb", -1), st.Tp, ws, n_split,
                                         pad_len=st.pad_len, d_past=d_past, merge_in_launch=bufs.get("attn_merge", False))
            else:
                # queries leave the RoPE kernel multiplied by scale * log2(e) (before their one rounding to bf16, as
                # phi.py:454 scales q before the product): the prefill attention's softmax is then the exponential alone
                if not fused_qkv:
                    ops.rope_kv_append(qkv, st.cos, st.sin, q, st.k[i], st.v[i], B, L, nh, nkv, hd, past, st.Tp, True, st.T, 1,
                                       q_scale=scale * ops.Q_PRESCALE)
                ops.attention(q, o, B, L, nh, nkv, hd, scale, True, past=past, k_past=st.k[i], v_past=st.v[i],
                              past_t=st.Tp, pad_len=st.pad_len, new_is_cache=True, q_prescaled=True)
            if mlx4_first:                                      # this layer attended on the exact keys (phi.py:533); from now on: the codes
                self._mlx4_quantize(st, i, L, qkv, nh)          # (keys from their exact fp32 values: recomputed from `qkv`)
            if last_only and i == cfg.num_hidden_layers - 1 and L > 1:
                o = o.view(B, L, -1)[:, -1].contiguous()
                x = x.view(B, L, -1)[:, -1].contiguous()
                a, h = a[:B], h[:B]
            # Short prompts and decode batches (9 .. 256 rows): o_proj and down_proj run as K slices, and the launch that adds the slices also writes the
            # RMSNorm of the new residual stream into `h` -- the next projection's input (ops.gemm_resid_norm; bit-identical to the
            # two launches it replaces).
            normed = False
            if not (bufs.get("fuse_o", False) and L <= ops.L.DECODE_MAX_L and n_beam == 1):
                normed = self._proj_resid_norm(o, p + "self_attn.o_proj.weight", x, w[p + "post_attention_layernorm.weight"], h)
                if not normed:
                    self._proj(o, p + "self_attn.o_proj.weight", EPI_RESID_BF16, resid=x, out=x)
            if normed:
                self._proj(h, p + "mlp.gate_up_proj.weight", EPI_SILU_MUL, out=a)
            else:
                self._proj(x, p + "mlp.gate_up_proj.weight", EPI_SILU_MUL, norm_w=w[p + "post_attention_layernorm.weight"], out=a, h=h)
            normed_in = False
            if i + 1 < cfg.num_hidden_layers:
                normed_in = self._proj_resid_norm(a, p + "mlp.down_proj.weight", x, w[f"model.layers.{i + 1}.input_layernorm.weight"], h)
            if not normed_in:
                self._proj(a, p + "mlp.down_proj.weight", EPI_RESID_BF16, resid=x, out=x)
            if self.hidden_hook is not None:                     # diagnostics only (tools/precision_decomp.py); never set
                self.hidden_hook(i, x, B, x.shape[0] // B)       # while a decode graph is captured
        if mlx4_first:
            st.mlx4_tokens = L
        return x

    def _mlx4_quantize(self, st, i, L, qkv, nh):
        if st.k4.shape[3] != L:                                 # (first call shorter than the state was sized for)
            nl, B, nkv, _, g, _ = st.k4.shape
            st.k4 = torch.zeros((nl, B, nkv, L, g, 4), dtype=I32, device=self.device)
            st.v4 = torch.zeros_like(st.k4)
            st.k_sb = torch.zeros((nl, B, nkv, L, g, 2), dtype=F32, device=self.device)
            st.v_sb = torch.zeros_like(st.k_sb)
        ops.kv_quantize_mlx4(st.k[i], st.v[i], st.k4[i], st.v4[i], st.k_sb[i], st.v_sb[i], L, qkv=qkv, cos_t=st.cos, sin_t=st.sin, nh=nh,
                             past=0, tab_t=st.T, tab_div=1)

    # ------------------------------------------------------------------ graph-replayed greedy decode step
    def _build_decode_graph(self, st):
        """Capture ONE greedy decode step (embed -> 32 layers -> norm+lm_head -> argmax ->
        bookkeeping) as a hipGraph.  All loop state lives in HBM: `tok` (next input ids),
        `d_past` (cache length), `d_step`, `history` -- so replays need no host input and
        a token costs one graph launch instead of ~170 kernel launches."""
        cfg, w, B, dev = self.cfg, self.w, st.B, self.device
        g = dict(tok=torch.zeros((B,), dtype=I32, device=dev), d_past=torch.zeros((1,), dtype=I32, device=dev),
                 d_step=torch.zeros((1,), dtype=I32, device=dev),
                 # the step's tokens land in PINNED HOST memory straight from `k_step_end` (one 4-byte store per row and step):
                 # the host loop reads them after the step's event, no D2H copy node sits between two replays (api.greedy_loop)
                 history=torch.zeros((B, max(1, st.max_tokens) + 1), dtype=I32).pin_memory(),
                 x=torch.empty((B, cfg.hidden_size), dtype=BF16, device=dev),
                 logits=torch.empty((B, cfg.vocab_size), dtype=BF16, device=dev),
                 next_tok=torch.zeros((B,), dtype=I32, device=dev), ticket=torch.zeros((1,), dtype=I32, device=dev))
        bufs = self._alloc_bufs(B, 1)
        self._split_plan(bufs, B, 1, st.Tp, st.quantized, serving=getattr(st, "serving", False) or self.serving)   # one split per tile of CAPACITY
        # the cache length only grows under a captured step (greedy_step rebuilds the graph if it ever finds it below this); a
        # slot state's column moves both ways (engine.py), so it gets no bound
        bufs["past_lb"] = -1 if getattr(st, "slots", False) else int(st.offset)
        if not getattr(st, "slots", False):
            self._plan_fused_oproj(bufs, B, 1, st.Tp, st.quantized)
        bufs["rope_cos"] = torch.empty((B, 1, self.hd // 2), dtype=F32, device=dev)
        bufs["rope_sin"] = torch.empty_like(bufs["rope_cos"])
        g["bufs"] = bufs

        g["amax_ws"] = torch.zeros((ops.L.GEMV_STEP_WS_BYTES // 4,), dtype=F32, device=dev)   # (its arrival counters start at zero)

        def step():
            # (round 6) the step's two ends have no launch of their own where the library folds them into the first / last projection
            # (B = 1 on bf16, e4m3 or 4-bit weights: ops.gemv_step_begin / gemv_step_end; 129 launches per step instead of 131)
            self._layers(g["x"], st, B, 1, 0, 1, bufs=bufs, d_past=g["d_past"],
                         step_begin=dict(tok=g["tok"], table=w["model.embed_tokens.weight"], cos_o=bufs["rope_cos"], sin_o=bufs["rope_sin"]))
            head = "lm_head.weight"
            w_fold = w.get(head) if head in w else (self.w8.get(head) or self.w4.get(head))   # bf16, (e4m3, row scales) or (4-bit, scale | bias)
            if not (w_fold is not None and head not in self.adapters and os.environ.get("P3V_STEP_FOLD", "1") != "0"
                    and ops.gemv_step_end(g["x"], w_fold, w["model.norm.weight"], cfg.rms_norm_eps, g["logits"], g["next_tok"], g["tok"],
                                          g["history"], g["d_step"], g["d_past"], g["ticket"], g["amax_ws"])):
                self._proj(g["x"], head, norm_w=w["model.norm.weight"], out=g["logits"], h=bufs["h"])   # (h: the norm's output
                ops.step_end(g["logits"], g["next_tok"], g["tok"], g["history"], g["d_step"], g["d_past"], g["ticket"])
        g["d_past"].fill_(st.offset)
        g["gemm_ws"] = {}                                        # B > 16 rows: the projections are split-K GEMMs; their workspace
        with ops.owned_gemm_workspace(g["gemm_ws"], frozen=False):   # belongs to the graph (sized here, baked in below)
            step()                                               # warm-up run (sets func attributes, pages code in)
        torch.cuda.synchronize()
        graph = ops.Graph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.owned_gemm_workspace(g["gemm_ws"], frozen=True):
            graph.begin()
            step()
            graph.end()
        torch.cuda.current_stream().wait_stream(side)
        g["d_step"].zero_()                                      # the warm-up run above counted as a step
        g["graph"] = graph
        return g

    @_on_device
    def greedy_prefill(self, max_tokens, **inputs):
        """Prefill + first greedy token: `logits, cache = model(**inputs, max_tokens); argmax(logits[:, -1])`
        (reference phi_3_vision_mlx.py:385-386).  Returns (token [B,1] int32 on the device, cache)."""
        logits, cache = self(**inputs, max_tokens=max_tokens)
        return ops.argmax(logits[:, -1, :].contiguous())[:, None], cache

    @_on_device
    def greedy_step(self, token, cache):
        """One greedy decode step through the captured graph.  Equivalent to
        `logits, cache = model(input_ids=token, cache=cache); next = argmax(logits[:, -1])`
        (reference phi_3_vision_mlx.py:391-392).  Returns (logits [B,1,V], next_token [B,1])
        -- views of persistent buffers, valid until the next call."""
        st = cache[0].state
        if st.offset + 1 > st.T:
            raise ValueError(f"KV cache overflow: {st.offset}+1 > {st.T} (prompt + max_tokens)")
        if st.epoch != self.epoch:                              # adapters changed since the capture: stale pointers
            st.graphs.clear()
            st.epoch = self.epoch
        g = st.graphs.get("greedy")
        if g is not None and st.offset < g["bufs"].get("past_lb", -1):   # the cache was rewound below the captured lower bound
            g = None
        if g is None:
            g = st.graphs["greedy"] = self._build_decode_graph(st)
            g["host_tok"] = None
        if g["host_tok"] is None or token is not g["host_tok"]:
            g["tok"].copy_(token.reshape(-1).to(self.device, I32))   # first step / caller-chosen token
        g["d_past"].fill_(st.offset) if g.get("synced_offset") != st.offset else None
        g["graph"].launch()
        g["n_replays"] = g.get("n_replays", 0) + 1              # replay r wrote its token to history[:, r - 1] (while it fits)
        st.offset += 1
        g["synced_offset"] = st.offset
        g["host_tok"] = g["next_tok"].view(-1, 1)
        return g["logits"].view(st.B, 1, -1), g["host_tok"]

    @_on_device
    def __call__(self, input_ids, pixel_values=None, image_sizes=None, positions=None, cache=None, pids=None, mask=None,
                 max_tokens=0, advance_offset=None, n_beam=1, full_logits=None):
        """Phi3ForCausalLM.__call__ (phi.py:606-608) + Phi3F.__call__ (phi.py:576-592).

        Returns (logits bf16 [B, L', V], cache).  On a prefill call (cache is None)
        only the last position is projected through lm_head (L' = 1) unless
        `full_logits=True`: every caller in the reference reads `logits[:, -1]`
        of a prefill only; cached calls return all L rows."""
        cfg, w = self.cfg, self.w
        if torch.is_tensor(input_ids):
            ids = input_ids.to(self.device, I32)
        else:
            ids = torch.as_tensor(np.asarray(input_ids).astype(np.int32)).to(self.device)
        if ids.dim() == 1:
            ids = ids[None]
        ids = ids.contiguous()
        B, L = ids.shape
        H = cfg.hidden_size
        prefill = cache is None
        if prefill and pixel_values is None and B == 1 and pids is None and mask is None and n_beam == 1 and not full_logits \
                and advance_offset is None:
            got = self._prefill_captured(ids, L, max_tokens)
            if got is not None:
                return got
        if prefill:
            # (before the vision tower is enqueued: the host is launch-bound through the ViT's ~280 launches, so whatever it does between
            #  the tower and the decoder -- cache allocation, its zero fill, the rotation tables -- would show as idle GPU time there)
            st = self._new_state(B, L, max_tokens, pids, mask)
            cache = [LayerCache(st, i) for i in range(cfg.num_hidden_layers)]
        else:
            st = cache[0].state
        x = ops.embed_gather(ids.view(-1), w["model.embed_tokens.weight"])
        if pixel_values is not None and self.vision:
            self.vision_embed(x, pixel_values, image_sizes, positions, L)
        past = st.offset
        if n_beam == 1 and past + L > st.T:
            raise ValueError(f"KV cache overflow: {past}+{L} > {st.T} (prompt + max_tokens)")
        if full_logits is None:
            full_logits = not prefill
        x = self._layers(x, st, B, L, past, n_beam, last_only=not full_logits)
        if n_beam == 1:
            st.offset = past + L                                 # KVCache.__call__ auto-advance (phi.py:544-547)
        if advance_offset is not None:
            st.offset = past + advance_offset                    # phi.py:589-591
        if not full_logits and x.shape[0] != B:
            x = x.view(B, L, H)[:, -1, :].contiguous()
        logits = self._proj(x, "lm_head.weight", norm_w=w["model.norm.weight"])
        return logits.view(B, -1, cfg.vocab_size), cache

    # ------------------------------------------------------------------ captured prefill of short text prompts
    PREFILL_GRAPH_MAX_S = 512
    PREFILL_GRAPH_ENTRIES = 4
    PREFILL_GRAPH_MAX_CACHE_BYTES = 2 << 30                       # an entry pins its (S + max_tokens) KV cache: 393 KB per token at full size

    def _prefill_captured(self, ids, S, max_tokens):
        """Short text prompts (B = 1, 17 .. 512 tokens) are launch-bound: ~390 launches of a few microseconds each, every one paid
        for in Python + ctypes time (BASELINE config 1: 4.5 ms of wall time for 3.6 ms of kernels).  A (length, max_tokens) pair seen
        for the SECOND time is captured as ONE hipGraph over buffers the entry owns -- ids, activations, rotation tables, the KV
        cache itself -- and later prompts of that geometry cost one 0.5 KB copy + one graph launch.  The entry's cache is LEASED to
        the caller: every call gets its own CacheState object (a shallow copy of the entry's: same tensors, same decode graphs, its
        own offset) and the lease IS that object -- while it is alive, through the returned cache list or through a kept
        `cache[i].state`, a new prompt of the same geometry takes the eager path, so a caller that keeps two caches gets two caches
        (the reference's contract).  An entry whose caller died in an unrecovered failed step is dropped (`CacheState.mark_dirty`),
        and entries whose cache would exceed PREFILL_GRAPH_MAX_CACHE_BYTES are not made.  Bit-identical to the eager path (same
        kernels, same order).  P3V_PREFILL_GRAPH=0 switches it off.  Returns None when not applicable."""
        cfg = self.cfg
        if (os.environ.get("P3V_PREFILL_GRAPH", "1") == "0" or S <= ops.L.DECODE_MAX_L or S > self.PREFILL_GRAPH_MAX_S or max_tokens < 1
                or getattr(cfg, "use_quantized_cache", False) or self.adapters or self.hidden_hook is not None or self.w8):
            return None                                           # (4-bit weights are captured too, round 6: their prefill is the bf16 one + a
                                                                  #  dequantise launch per projection into the model's own scratch)
        if 4 * cfg.num_hidden_layers * cfg.num_key_value_heads * self.hd * (S + max_tokens + 256) > self.PREFILL_GRAPH_MAX_CACHE_BYTES:
            return None
        # the graph bakes in the rotation tables and the weight pointers: both are part of the key (tests mutate `cfg` and swap weight
        # tensors on a live model; in-place weight updates are seen by the graph as they are by every launch)
        nl = cfg.num_hidden_layers - 1
        def wptr(k):
            return self.w[k].data_ptr() if k in self.w else self.w4[k][0].data_ptr()
        key = (S, max_tokens, rope_scaling_factor(cfg), float(cfg.rope_theta), cfg.original_max_position_embeddings,
               self.w["model.embed_tokens.weight"].data_ptr(), wptr("lm_head.weight"),
               wptr("model.layers.0.self_attn.qkv_proj.weight"), wptr(f"model.layers.{nl}.mlp.down_proj.weight"))
        e = self._prefill_graphs.get(key)
        if e is None:
            if self._prefill_seen.get(key, 0) < 1:                # first sighting: eager (one-off lengths never pay for a capture)
                if len(self._prefill_seen) > 64:
                    self._prefill_seen.clear()
                self._prefill_seen[key] = 1
                return None
            if len(self._prefill_graphs) >= self.PREFILL_GRAPH_ENTRIES:
                self._prefill_graphs.pop(next(iter(self._prefill_graphs)))
            e = self._prefill_graphs[key] = self._build_prefill_graph(S, max_tokens)
        if e["lease"] is not None and e["lease"]() is not None:    # the previous caller still holds this entry's cache
            return None
        if e["st"].shared["dirty"]:                               # its last caller died in a failed step: NaN rows, half-armed buffers
            del self._prefill_graphs[key]
            self._prefill_seen[key] = 1                           # (the next prompt of this geometry captures afresh)
            return None
        st = copy.copy(e["st"])                                   # this call's state: the entry's tensors and graphs, its own offset
        self._states.add(st)
        e["ids"].copy_(ids.view(-1), non_blocking=True)
        st.offset = 0
        g = st.graphs.get("greedy")
        if g is not None:                                         # the decode graph of this cache geometry stays valid: same buffers
            g["n_replays"], g["host_tok"], g["synced_offset"] = 0, None, None
            g["d_step"].zero_()
        e["graph"].launch()
        st.offset = S
        e["lease"] = weakref.ref(st)
        return e["logits"].clone().view(1, 1, cfg.vocab_size), [LayerCache(st, i) for i in range(cfg.num_hidden_layers)]

    def _build_prefill_graph(self, S, max_tokens):
        cfg, w, dev = self.cfg, self.w, self.device
        st = self._new_state(1, S, max_tokens, None, None)
        e = dict(st=st, lease=None, ids=torch.zeros((S,), dtype=I32, device=dev), x=torch.empty((S, cfg.hidden_size), dtype=BF16, device=dev),
                 logits=torch.empty((1, cfg.vocab_size), dtype=BF16, device=dev), bufs=self._alloc_bufs(1, S), ws={})

        def run():
            st.offset = 0
            ops.embed_gather(e["ids"], w["model.embed_tokens.weight"], out=e["x"])
            xl = self._layers(e["x"], st, 1, S, 0, 1, bufs=e["bufs"], last_only=True)
            self._proj(xl.view(1, -1), "lm_head.weight", norm_w=w["model.norm.weight"], out=e["logits"])
        with ops.owned_gemm_workspace(e["ws"], frozen=False):
            run()                                                # warm-up (sizes the split-K workspace the graph owns)
        torch.cuda.synchronize()
        graph = ops.Graph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.owned_gemm_workspace(e["ws"], frozen=True):
            graph.begin()
            run()
            graph.end()
        torch.cuda.current_stream().wait_stream(side)
        e["graph"] = graph
        return e

    @property
    def layers(self):
        return range(self.cfg.num_hidden_layers)
