"""CPU ORACLE (test infrastructure only -- never imported by the product path).

A PyTorch-CPU restatement of the reference model `/root/reference/phi.py`,
op for op and with the reference's dtype flow:

  * weights are bf16; `Linear(bf16 x)` -> bf16 (fp32 accumulate, one rounding);
    `Linear(fp32 x)` -> fp32 (dtype promotion);
  * RoPE tables are fp32, so q,k are promoted to fp32, the KV cache is fp32,
    attention runs in fp32 and only the o_proj result is cast back to bf16
    (phi.py:451-460, 494-504, 543);
  * the vision tower and projector run on fp32 activations (phi.py:279, 309);
  * logits are bf16 (phi.py:608).

PARITY STATUS: **pinned to the reference's own code** (round 4).  The reference's arithmetic
lives in the un-vendored `mlx==0.15.0` wheel, which is not installable here, so
`tests/golden/ref_env.py` imports the reference's `phi.py` / `phi_3_vision_mlx.py` where they
lie over a functional stand-in for the MLX API (`tests/golden/mlx_shim.py`: documented MLX
semantics, no import of this file), `tests/golden/gen_golden_refmodel.py` runs the reference's
`_load`, `_generate`, `_choose_from`, `_constrain` through it and commits the outputs
(`tests/golden/ref_model_{tiny,full,wc}.*`), and `tests/test_refmodel.py` holds this oracle to
them: text paths bit for bit (every logit of every step, cache contents, every model call
of the constrain loops), image paths token-exact with logits inside 2^-7 of the row's maximum.
What stays unpinned is the bits of MLX's Metal kernels (reduction order): nobody can run them
here.  The NumPy image preprocessing is pinned separately (tests/golden/gen_golden_ref.py).
Each function cites the phi.py lines it follows.

MLX op semantics restated from the MLX documentation (not in-tree):
  nn.gelu_fast_approx(x) = x*sigmoid(1.702x); nn.GELU() = exact erf GELU;
  nn.RMSNorm = x*rsqrt(mean(x^2)+eps)*w with fp32 accumulation;
  nn.LayerNorm eps=1e-5, biased variance; softmax/log_softmax fp32 internal;
  bf16 (x) fp32 -> fp32 promotion; argmax returns the first maximum;
  mx.repeat(axis=0) == torch.repeat_interleave.

Quirk handling: Q7 -- fully masked (left-pad) query rows give
NaN in IEEE arithmetic in the reference; the oracle defines them as 0 output and
gives pad keys exactly zero weight; only valid rows are ever compared.
"""
import math

import torch

BF16 = torch.bfloat16
F32 = torch.float32


def _linear(x, w, b=None):
    """nn.Linear with MLX promotion: bf16 x -> bf16 out; fp32 x -> fp32 out."""
    y = x.to(F32) @ w.to(F32).t()
    if b is not None:
        y = y + b.to(F32)
    return y.to(BF16) if x.dtype == BF16 else y


def rms_norm(x, w, eps):
    """nn.RMSNorm (phi.py:478-479,571) = mx.fast.rms_norm: fp32 statistics, the normalised row is rounded to x.dtype and
    THEN multiplied by the weight in x.dtype -- `w * astype(x * rsqrt(mean(x^2) + eps), T)`, two roundings for bf16 x."""
    xf = x.to(F32)
    n = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)
    return n * w.to(x.dtype)


def layer_norm(x, w, b, eps):
    """nn.LayerNorm (phi.py:165,167,212)."""
    xf = x.to(F32)
    mu = xf.mean(-1, keepdim=True)
    var = (xf - mu).pow(2).mean(-1, keepdim=True)
    return ((xf - mu) * torch.rsqrt(var + eps) * w.to(F32) + b.to(F32)).to(x.dtype)


def rotate_half(x, cos, sin):
    """phi.py:418-423 -- half-split convention; bf16*fp32 promotes to fp32."""
    mid = x.shape[-1] // 2
    x1, x2 = x[..., :mid], x[..., mid:]
    xf = x.to(F32)
    return xf * cos + torch.cat([-x2, x1], dim=-1).to(F32) * sin


def su_rope_tables(cfg, L_all, pids):
    """SuRoPE.__init__ (phi.py:487-504). Returns cos, sin [B,1,L_all,dim] fp32."""
    dim = cfg.hidden_size // cfg.num_attention_heads
    scaling_factor = math.sqrt(1 + math.log(cfg.max_position_embeddings / cfg.original_max_position_embeddings)
                               / math.log(cfg.original_max_position_embeddings))
    su = cfg.rope_scaling["long_factor"] if L_all > cfg.original_max_position_embeddings else cfg.rope_scaling["short_factor"]
    if pids is None:
        position_ids = torch.arange(L_all, dtype=F32)[None]
    else:
        pids = torch.as_tensor(pids)
        ext = pids[:, -1][:, None].to(F32) + 1 + torch.arange(L_all - pids.shape[1], dtype=F32)[None, :]
        position_ids = torch.cat([pids.to(F32), ext], dim=1)
    inv_freq = 1.0 / (torch.tensor(su, dtype=F32) * (torch.tensor(float(cfg.rope_theta), dtype=F32)
                                                    ** (torch.arange(0, dim, 2, dtype=F32) / dim)))
    freqs = position_ids[:, :, None] * inv_freq[None, None, :]        # [B, L, dim/2]
    emb = torch.cat([freqs, freqs], dim=-1)
    cos = (torch.cos(emb) * scaling_factor)[:, None]
    sin = (torch.sin(emb) * scaling_factor)[:, None]
    return cos, sin


def mx_quantize(w, group_size=32, bits=4):
    """mx.quantize of the pinned mlx 0.15.0 (restated here: the oracle imports nothing of the product): per group of `group_size`
    values w ~ scale * q + bias with q in 0 .. 2^bits - 1; the end of the range with the larger magnitude is represented exactly
    (it becomes the bias, the scale takes the sign that reaches the other end).  In that release the function is a composite of
    array primitives in w's dtype, so for the reference's bf16 VALUES (phi.py:443-449: only q and k are promoted to fp32 by the
    rotation) every intermediate rounds to bf16 -- `r` below -- while its fp32 KEYS see plain fp32 arithmetic.
    -> (codes int64 [N, K], scales [N, K / group], biases) with scales / biases in w's dtype."""
    N, K = w.shape
    dt = w.dtype
    r = (lambda t: t) if dt == torch.float32 else (lambda t: t.to(dt).float())
    g = w.float().reshape(N, K // group_size, group_size)
    n_bins = float((1 << bits) - 1)
    w_max, w_min = g.amax(-1), g.amin(-1)
    side = w_min.abs() > w_max.abs()
    scales = torch.maximum(r(r(w_max - w_min) / n_bins), r(torch.tensor(1e-7)))
    scales = torch.where(side, scales, -scales)
    edge = torch.where(side, w_min, w_max)
    q0 = torch.round(r(edge / scales))
    scales = torch.where(q0 != 0, r(edge / q0), scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    q = torch.round(r(r(g - biases[..., None]) / scales[..., None])).clamp(0, n_bins).to(torch.int64)
    return q.reshape(N, K), scales.to(dt), biases.to(dt)


def mx_dequantize(q, scales, biases, group_size=32):
    """mx.dequantize of the same release: multiply(codes, scales) then add(., biases), two primitives that each round to the
    scales' dtype."""
    N, K = q.shape
    dt = scales.dtype
    prod = (q.float().reshape(N, K // group_size, group_size) * scales.float()[..., None]).to(dt)
    return (prod.float() + biases.float()[..., None]).to(dt).reshape(N, K)


class OracleKVCache:
    """KVCache (phi.py:509-548): the plain path + beam view, and (round 5) the reference's quantised variant
    (`use_quantized_cache`: phi.py:528-540 -- mx.quantize(group 32, 4 bits) of the FIRST call's keys / values, later tokens kept
    unquantised, every later call attends on the dequantised prompt + those)."""

    def __init__(self, cfg, B, S, max_tokens):
        self.max_tokens = max_tokens
        self.offset = 0
        self.shape = (2, B, cfg.num_key_value_heads, S + max_tokens, cfg.hidden_size // cfg.num_key_value_heads)
        self.kv = None
        self.use_quantized_cache = bool(getattr(cfg, "use_quantized_cache", False)) and getattr(cfg, "cache_format", "int8") == "mlx4"
        self.keys, self.values = [], []

    def __call__(self, keys, values, n_beam):
        if self.max_tokens < 1:
            return keys, values
        if n_beam > 1:
            if self.use_quantized_cache:
                raise NotImplementedError("Beam Search is not yet compatible with Quantized Cache")
            kv = self.kv[:, :, :, :self.offset, :].repeat_interleave(n_beam, dim=1)
            return torch.cat([kv[0], keys.to(kv.dtype)], dim=-2), torch.cat([kv[1], values.to(kv.dtype)], dim=-2)
        if self.use_quantized_cache:                             # phi.py:528-540
            self.offset += values.shape[2]
            _, B, N, _, D = self.shape
            if self.kv is None:
                self.kv = (mx_quantize(keys.reshape(B * N, -1)), mx_quantize(values.reshape(B * N, -1)))
                return keys, values
            self.keys.append(keys)
            self.values.append(values)
            k_cache = mx_dequantize(*self.kv[0]).reshape(B, N, -1, D)
            v_cache = mx_dequantize(*self.kv[1]).reshape(B, N, -1, D)
            return torch.cat([k_cache] + self.keys, dim=2), torch.cat([v_cache] + self.values, dim=2)
        if self.kv is None:
            self.kv = torch.zeros(self.shape, dtype=keys.dtype)
        new_offset = self.offset + keys.shape[2]
        self.kv[0, :, :, self.offset:new_offset, :] = keys
        self.kv[1, :, :, self.offset:new_offset, :] = values.to(self.kv.dtype)
        self.offset = new_offset
        return self.kv[0, :, :, :new_offset, :], self.kv[1, :, :, :new_offset, :]


class OracleMask4D:
    """Mask4D (phi.py:550-563); kept as a boolean 'allowed' tensor."""

    def __init__(self, L_all, mask):
        allowed = torch.tril(torch.ones(L_all, L_all, dtype=torch.bool))[None, None]
        if mask is not None:
            mask = torch.as_tensor(mask)
            m = torch.nn.functional.pad(mask, (0, L_all - mask.shape[-1]), value=1)
            m = m[:, None, None, :]
            allowed = allowed & ((m * m.transpose(2, 3)) == 1)
        self.allowed = allowed

    def __call__(self, past_L, L):
        return self.allowed[:, :, past_L:L + past_L, :L + past_L]


def masked_softmax(w, allowed):
    """softmax(w + mask) with Q7 semantics for fully masked rows (-> zeros)."""
    w = w.masked_fill(~allowed, float("-inf"))
    m = w.max(dim=-1, keepdim=True).values
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    e = torch.exp(w - m)
    s = e.sum(-1, keepdim=True)
    return torch.where(s > 0, e / s.clamp_min(1e-38), torch.zeros_like(e))


class OraclePhi3V:
    """Phi3VForCausalLM / Phi3ForCausalLM (phi.py:565-617) on CPU."""

    def __init__(self, cfg, weights, cache_fp32=False, adapters=None):
        """adapters: {"model.layers.<i>.<target>.weight": (lora_a [in,r] f32, lora_b [r,out] f32, scale)} --
        the LoRA layers `_linear_to_lora_layers` would install (phi_3_vision_mlx.py:234-245)."""
        self.cfg = cfg
        self.w = weights
        self.adapters = adapters or {}
        self._f32 = {} if cache_fp32 else None
        self.vision = cfg.architectures[0].startswith("Phi3V")
        self._masker = None
        self._roper = None

    # -- weight access ---------------------------------------------------
    def W(self, name):
        """bf16 weight; with cache_fp32 an fp32 copy (same values) is kept to
        avoid re-converting on every call."""
        if self._f32 is None or name == "model.embed_tokens.weight":
            return self.w[name]
        t = self._f32.get(name)
        if t is None:
            t = self._f32[name] = self.w[name].to(F32)
        return t

    def proj(self, x, name):
        """nn.Linear, or LoRALinear.__call__ (phi.py:129-133) when the projection carries an adapter:
        y = linear(x); z = (x @ lora_a) @ lora_b  (fp32 by promotion); (y + scale * z).astype(x.dtype)."""
        y = _linear(x, self.W(name))
        ad = self.adapters.get(name)
        if ad is None:
            return y
        a, b, scale = ad
        z = (x.to(F32) @ a.to(F32)) @ b.to(F32)
        return (y.to(F32) + scale * z).to(x.dtype)

    # -- vision tower (phi.py:135-221) ------------------------------------
    def clip_embeddings(self, x_nchw):
        """ClipEmbeddings (phi.py:197-206); x [N,3,336,336] fp32."""
        v = "model.vision_embed_tokens.img_processor.vision_model.embeddings."
        wpe = self.W(v + "patch_embedding.weight").to(F32)
        P = wpe.shape[-1]
        pe = torch.nn.functional.conv2d(x_nchw, wpe, stride=P)          # [N,D,24,24]
        pe = pe.flatten(2).transpose(1, 2)                               # [N,576,D]
        cls = self.W(v + "class_embedding").to(F32)[None, None].expand(pe.shape[0], 1, -1)
        emb = torch.cat([cls, pe], dim=1)
        return emb + self.W(v + "position_embedding.weight").to(F32)[None]

    def clip_layer(self, x, j):
        """ClipEncoderLayer (phi.py:169-171) with ClipAttention (:145-149), ClipMLP (:158-159)."""
        c = self.cfg.clip
        q = f"model.vision_embed_tokens.img_processor.vision_model.encoder.layers.{j}."
        nh = c["num_attention_heads"]
        B, L, D = x.shape
        h = layer_norm(x, self.W(q + "layer_norm1.weight"), self.W(q + "layer_norm1.bias"), c["layer_norm_eps"])
        qs, ks, vs = (_linear(h, self.W(q + f"self_attn.{n}.weight"), self.W(q + f"self_attn.{n}.bias"))
                      .reshape(B, L, nh, -1).transpose(1, 2) for n in ("q_proj", "k_proj", "v_proj"))
        scale = (D // nh) ** -0.5
        a = torch.softmax((qs * scale) @ ks.transpose(-1, -2), dim=-1) @ vs
        a = a.transpose(1, 2).reshape(B, L, D)
        x = x + _linear(a, self.W(q + "self_attn.out_proj.weight"), self.W(q + "self_attn.out_proj.bias"))
        h = layer_norm(x, self.W(q + "layer_norm2.weight"), self.W(q + "layer_norm2.bias"), c["layer_norm_eps"])
        h = _linear(h, self.W(q + "mlp.fc1.weight"), self.W(q + "mlp.fc1.bias"))
        h = h * torch.sigmoid(1.702 * h)                                 # nn.gelu_fast_approx
        return x + _linear(h, self.W(q + "mlp.fc2.weight"), self.W(q + "mlp.fc2.bias"))

    def clip_model(self, x_nchw):
        """ClipModel.__call__ (phi.py:216-221): all layers but the last, drop CLS, no post-LN."""
        v = "model.vision_embed_tokens.img_processor.vision_model."
        x = self.clip_embeddings(x_nchw)
        x = layer_norm(x, self.W(v + "pre_layrnorm.weight"), self.W(v + "pre_layrnorm.bias"), 1e-5)
        for j in range(self.cfg.clip["num_hidden_layers"] - 1):
            x = self.clip_layer(x, j)
        return x[:, 1:]

    def image_embedding(self, txt_embeds, pixel_values, image_sizes, positions, return_parts=False):
        """Phi3ImageEmbedding.__call__ (phi.py:393-416)."""
        e = "model.vision_embed_tokens."
        pv = torch.as_tensor(pixel_values).cpu().to(F32)                 # mx.array(f64) -> fp32
        B = pv.shape[0]
        img_sizes = (torch.as_tensor(image_sizes) // 336).tolist()
        positions = torch.as_tensor(positions).tolist()
        feats = self.clip_model(pv.reshape(-1, *pv.shape[2:]))
        feats = feats.reshape(B, -1, *feats.shape[1:])                   # [B,17,576,C]
        C, H = self.cfg.img_processor["image_dim_out"], int(feats.shape[2] ** 0.5)
        sub_GN, glb_GN = self.W(e + "sub_GN").to(F32), self.W(e + "glb_GN").to(F32)

        def rc(img, shape, tile_shape):
            t = img.reshape(shape).permute(0, 1, 3, 2, 4, 5).reshape(tile_shape)
            return torch.cat([t, sub_GN.expand(1, tile_shape[1], 1, -1)], dim=2).reshape(1, -1, 4 * C)

        outs, lens, merged = [], [], []
        for b in range(B):
            h, w = img_sizes[b]
            B_ = h * w
            glb = rc(feats[b, :1], (1, H // 2, 2, H // 2, 2, C), (1, H // 2, H // 2, 4 * C))
            sub = rc(feats[b, 1:B_ + 1], (B_, H // 2, 2, H // 2, 2, C), (1, h * 12, w * 12, 4 * C))
            x = torch.cat([sub, glb_GN, glb], dim=1)
            merged.append(x)
            x = _linear(x, self.W(e + "img_projection.0.weight"), self.W(e + "img_projection.0.bias"))
            x = torch.nn.functional.gelu(x)                              # nn.GELU(): exact erf
            x = _linear(x, self.W(e + "img_projection.2.weight"), self.W(e + "img_projection.2.bias"))
            outs.append(x)
            lens.append(int((h * w + 1) * 144 + 1 + (h + 1) * 12))
        idx = 0
        for i, cnt in enumerate(lens):
            r, c0 = positions[idx]
            txt_embeds[r, c0:c0 + cnt] = outs[i][0].to(txt_embeds.dtype)
            idx += cnt
        if return_parts:
            return txt_embeds, feats, merged, outs
        return txt_embeds

    # -- decoder (phi.py:425-485) ------------------------------------------
    def attention(self, x, i, cache, cos, sin, allowed, n_beam):
        """Phi3Attention.__call__ (phi.py:440-460)."""
        cfg = self.cfg
        p = f"model.layers.{i}.self_attn."
        nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        hd = cfg.hidden_size // nh
        B, L, _ = x.shape
        qkv = self.proj(x, p + "qkv_proj.weight")
        q, k, v = torch.split(qkv, [nh * hd, nkv * hd, nkv * hd], dim=-1)
        q = q.reshape(B, L, nh, -1).transpose(1, 2)
        k = k.reshape(B, L, nkv, -1).transpose(1, 2)
        v = v.reshape(B, L, nkv, -1).transpose(1, 2)
        if n_beam > 1:
            sin = sin.repeat_interleave(n_beam, dim=0)
            cos = cos.repeat_interleave(n_beam, dim=0)
            allowed = allowed.repeat_interleave(n_beam, dim=0)
        q = rotate_half(q, cos, sin)
        k = rotate_half(k, cos, sin)
        k, v = cache(k, v, n_beam)
        w = (q * (hd ** -0.5)) @ k.transpose(-1, -2)
        w = masked_softmax(w, allowed)
        o = w @ v.to(F32)
        o = o.transpose(1, 2).reshape(B, L, -1)
        return self.proj(o, p + "o_proj.weight").to(qkv.dtype)

    def mlp(self, x, i):
        """Phi3MLP.__call__ (phi.py:468-471); nn.silu on bf16, per-op bf16 rounding."""
        p = f"model.layers.{i}.mlp."
        y = self.proj(x, p + "gate_up_proj.weight")
        gate, up = torch.chunk(y, 2, dim=-1)
        act = gate * torch.sigmoid(gate)
        return self.proj(act * up, p + "down_proj.weight")

    def decoder_layer(self, x, i, cache, cos, sin, allowed, n_beam):
        """Phi3DecoderLayer.__call__ (phi.py:481-485)."""
        eps = self.cfg.rms_norm_eps
        p = f"model.layers.{i}."
        r = self.attention(rms_norm(x, self.W(p + "input_layernorm.weight"), eps), i, cache, cos, sin, allowed, n_beam)
        h = x + r
        r = self.mlp(rms_norm(h, self.W(p + "post_attention_layernorm.weight"), eps), i)
        return h + r

    def embed(self, input_ids):
        """nn.Embedding; negative ids (image slots) wrap in MLX -- clamp; the rows
        are overwritten by the image embeddings anyway (phi.py:414, Q6)."""
        ids = torch.as_tensor(input_ids).long().clamp_min(0)
        return self.W("model.embed_tokens.weight")[ids]

    def backbone(self, input_ids, pixel_values, image_sizes, positions, cache, pids, mask, max_tokens,
                 advance_offset, n_beam, hidden_hook=None):
        """Phi3F.__call__ (phi.py:576-592)."""
        cfg = self.cfg
        x = self.embed(input_ids).clone()
        if pixel_values is not None and self.vision:
            x = self.image_embedding(x, pixel_values, image_sizes, positions)
        if cache is None:
            cache = [OracleKVCache(cfg, x.shape[0], x.shape[1], max_tokens) for _ in range(cfg.num_hidden_layers)]
            self._masker = OracleMask4D(x.shape[1] + max_tokens, mask)
            self._roper = su_rope_tables(cfg, x.shape[1] + max_tokens, pids)
        past_L, new_L = cache[0].offset, x.shape[1]
        allowed = self._masker(past_L, new_L)
        cos, sin = (t[:, :, past_L:past_L + new_L, :] for t in self._roper)
        for i in range(cfg.num_hidden_layers):
            x = self.decoder_layer(x, i, cache[i], cos, sin, allowed, n_beam)
            if hidden_hook is not None:
                hidden_hook(i, x)
        if advance_offset is not None:
            for c in cache:
                c.offset = past_L + advance_offset
        return rms_norm(x, self.W("model.norm.weight"), cfg.rms_norm_eps), cache

    def __call__(self, input_ids, pixel_values=None, image_sizes=None, positions=None, cache=None, pids=None,
                 mask=None, max_tokens=0, advance_offset=None, n_beam=1, hidden_hook=None):
        """Phi3ForCausalLM.__call__ (phi.py:606-608): logits over ALL positions, bf16."""
        x, cache = self.backbone(input_ids, pixel_values, image_sizes, positions, cache, pids, mask, max_tokens,
                                 advance_offset, n_beam, hidden_hook)
        return _linear(x, self.W("lm_head.weight")), cache


# ---------------------------------------------------------------------------
# Decoding loops (host logic of reference phi_3_vision_mlx.py), restated on
# torch CPU tensors.  `model` is an OraclePhi3V; inputs are already-tokenised.
# ---------------------------------------------------------------------------
ID_EOS = 32007


def log_softmax(x):
    """nn.log_softmax(x) = x - mx.logsumexp(x, keepdims=True) (MLX's composite): the log-sum-exp is accumulated in fp32,
    ROUNDED to x.dtype, and the subtraction rounds once more -- for bf16 logits every entry of a row carries the same
    rounding of the normaliser (it cancels inside a row, not between rows)."""
    lse = torch.logsumexp(x.to(F32), dim=-1, keepdim=True).to(x.dtype)
    return x - lse


def mean_last(x):
    """mx.mean over the last axis (MLX's composite): sum (fp32 accumulate, rounded to x.dtype) * (1/n rounded to x.dtype)."""
    return _sum_last(x) * torch.tensor(1.0 / x.shape[-1], dtype=x.dtype)


def greedy_generate(model, dict_input, max_tokens, stop_on_eos=True):
    """`_generate` (phi_3_vision_mlx.py:376-400) without streaming/LogitStopper.
    Returns tokens [B, n_steps] (int64) and the per-step last-position logits."""
    mask, pids = dict_input.get("mask"), dict_input.get("pids")
    logits, cache = model(**dict_input, max_tokens=max_tokens)
    token = torch.argmax(logits[:, -1, :].to(F32), dim=-1)[:, None]
    toks, lgs = [token], [logits[:, -1, :]]
    eos_rows = torch.ones(token.shape[0])
    for _ in range(max_tokens - 1):
        logits, cache = model(input_ids=token, cache=cache, mask=mask, pids=pids)
        token = torch.argmax(logits[:, -1, :].to(F32), dim=-1)[:, None]
        toks.append(token)
        lgs.append(logits[:, -1, :])
        if stop_on_eos and (token == ID_EOS).any():                      # TokenStopper (:105-117)
            eos_rows = eos_rows * (token.squeeze(1) != ID_EOS)
            if eos_rows.sum() < 1:
                break
    return torch.cat(toks, dim=1), torch.stack(lgs, dim=1)


def choose_from(model, dict_input, options):
    """`_choose_from` core (phi_3_vision_mlx.py:473-477): index of best option per row."""
    logits, _ = model(**dict_input, max_tokens=0)
    lp = log_softmax(logits[:, -1, :])
    return torch.argmax(lp[:, torch.as_tensor(options).long()].to(F32), dim=-1).tolist()


def _already(a2, a1):
    """phi_3_vision_mlx.py:495-498 (returns 1 where the row does NOT yet end with a1)."""
    if a2.shape[1] < a1.shape[0]:
        return torch.ones(a2.shape[0])
    return (~torch.all(a2[:, -len(a1):] == a1, dim=1)).to(F32)


def top3_candidates(row_logits, n_beam=3):
    """Deterministic stand-in for mx.argpartition(-logits, kth=n)[:, :n] (Q9):
    the n largest, ordered by (-logit, index)."""
    lf = row_logits.to(F32)
    order = torch.sort(lf, dim=-1, descending=True, stable=True).indices
    return order[:, :n_beam]


def _sum_last(x):
    """sum over the last axis of a bf16/fp32 array: fp32 accumulate, result in x.dtype."""
    return x.to(F32).sum(-1).to(x.dtype)


def _div(x, n):
    return (x.to(F32) / n).to(x.dtype)


def constrain_one(model, dict_input, constraint, id_constraint, use_beam=False, log_norm=False, trace=None):
    """One (max_new, text) constraint of `_constrain` (phi_3_vision_mlx.py:537-601).
    Scores stay in the logits dtype (bf16) as in the reference; reductions
    accumulate in fp32 and round once (sum) then once more (divide).
    Returns (synth_sofar [B, *] token ids padded with ID_EOS, score_sofar [B]).

    trace (test infrastructure, not in the reference): a list that receives one (kind, margin, outcome) record per
    data-dependent DECISION of the loop -- argmax / top-3 picks (margin = gap to the runner-up) and score comparisons
    (margin = |a - b|), margins divided by max|logit| of the forward that produced them, outcome = the picked ids /
    booleans as a list -- so a parity test can walk two runs decision by decision and tell a near-tie flip from a bug."""
    scale_box = [1.0]

    def _note(kind, margin, outcome):
        if trace is not None:
            trace.append((kind, float(torch.as_tensor(margin).to(F32).min()) / scale_box[0], torch.as_tensor(outcome).reshape(-1).tolist()))

    def _gap(row_logits, k=1):
        """gap between the k-th and (k+1)-th largest entry of every row (min over rows)."""
        v = row_logits.to(F32).topk(k + 1, dim=-1).values
        return (v[..., k - 1] - v[..., k]).min()

    def _lsm(logits):
        scale_box[0] = max(float(logits.to(F32).abs().max()), 1e-6)
        return log_softmax(logits)

    def _log_mean(x):
        if log_norm:
            return _div(_sum_last(x), math.log(x.shape[-1]))
        return _div(_sum_last(x), x.shape[-1])

    idc = torch.as_tensor(id_constraint).long()
    Bn = torch.as_tensor(dict_input["input_ids"]).shape[0]
    synth_pad = torch.full((Bn, 1), ID_EOS, dtype=torch.long)
    ar = torch.arange

    def _get_beam(logits, cache, beam_idx=0, n_beam=3):
        token = torch.argmax(logits[:, beam_idx, :].to(F32), dim=-1)
        arg_beam = top3_candidates(logits[:, beam_idx, :], n_beam)
        _note("argmax", _gap(logits[:, beam_idx, :]), token)
        _note("top3_set", _gap(logits[:, beam_idx, :], n_beam), arg_beam.sort(dim=-1).values)
        beam = arg_beam.reshape(-1)[:, None]
        beam = torch.cat([beam, idc[None].expand(beam.shape[0], -1)], dim=-1)
        bl, _ = model(input_ids=beam, cache=cache, n_beam=n_beam, advance_offset=0)
        bl = _lsm(bl)
        s0 = logits[ar(arg_beam.shape[0])[:, None], beam_idx, arg_beam].reshape(-1)[:, None]
        s1 = bl[ar(bl.shape[0])[:, None], ar(beam.shape[1] - 1)[None, :], beam[:, 1:]]
        beam_score_all = torch.cat([s0, s1], dim=1)
        mean = mean_last(beam_score_all)                                   # `_beam_score.mean(axis=1)` (:513)
        amax = torch.argmax(mean.reshape(-1, n_beam).to(F32), dim=-1)
        _note("beam_pick", _gap(mean.reshape(-1, n_beam)), arg_beam[ar(amax.shape[0]), amax])
        beam_token = arg_beam[ar(amax.shape[0]), amax]
        beam_score = beam_score_all.reshape(logits.shape[0], n_beam, -1)[ar(amax.shape[0]), amax]
        return token, beam_token, beam_score

    logits, cache = model(**dict_input, max_tokens=constraint[0] + idc.shape[0] + 10)
    logits = _lsm(logits)
    score_0 = logits[:, -1, idc[0]]
    tiled = idc[None].expand(Bn, -1)
    lr, _ = model(input_ids=tiled, cache=cache, advance_offset=0)
    lr = log_softmax(lr)
    score_1 = lr[ar(Bn)[:, None], ar(tiled.shape[1] - 1)[None, :], tiled[:, 1:]]
    running_score = logits[:, -1, :].max(dim=-1).values[:, None]
    pre_score = _log_mean(torch.cat([score_0[:, None], score_1], dim=1))
    pre_synth = torch.cat([tiled, synth_pad], dim=1)
    if use_beam and constraint[0] > 0:
        token, beam_token, beam_score = _get_beam(logits, cache, -1)
        post_score = _log_mean(beam_score)
        post_synth = torch.cat([beam_token[:, None], tiled], dim=1)
        win = pre_score > post_score
        _note("pre_vs_post", (pre_score.to(F32) - post_score.to(F32)).abs(), win)
        score_sofar = torch.where(win, pre_score, post_score)
        synth_sofar = torch.where(win[:, None], pre_synth, post_synth)
    else:
        token = torch.argmax(logits[:, -1, :].to(F32), dim=-1)
        _note("argmax", _gap(logits[:, -1, :]), token)
        score_sofar, synth_sofar = pre_score, pre_synth
    token = token[:, None]
    tokens = []
    finished = torch.ones(Bn)
    for _ in range(constraint[0]):
        tokens.append(token)
        token_plus = torch.cat([token, tiled], dim=1)
        logits, cache = model(input_ids=token_plus, cache=cache, advance_offset=1)
        logits = _lsm(logits)
        g = logits[ar(Bn)[:, None], ar(logits.shape[1] - 1)[None, :], token_plus[:, 1:]]
        pre_score = _log_mean(torch.cat([running_score, g], dim=1))
        pre_synth = torch.cat(tokens + [tiled, synth_pad], dim=1)
        if use_beam:
            token, beam_token, beam_score = _get_beam(logits, cache)
            post_score = _log_mean(torch.cat([running_score, beam_score], dim=1))
            post_synth = torch.cat(tokens + [beam_token[:, None], tiled], dim=1)
            win = pre_score > post_score
            _note("pre_vs_post", (pre_score.to(F32) - post_score.to(F32)).abs(), win)
            score = torch.where(win, pre_score, post_score)
            synth = torch.where(win[:, None], pre_synth, post_synth)
        else:
            token = torch.argmax(logits[:, 0, :].to(F32), dim=-1)
            _note("argmax", _gap(logits[:, 0, :]), token)
            score, synth = pre_score, pre_synth
        synth_sofar = torch.cat([synth_sofar, synth_pad], dim=1)
        finished = finished * _already(torch.cat(tokens, dim=1), idc)
        upd = (score > score_sofar).to(F32) * finished
        _note("update", (score.to(F32) - score_sofar.to(F32)).abs(), upd)
        synth_sofar = torch.where(upd[:, None] > 0, synth, synth_sofar)
        score_sofar = torch.where(upd > 0, score, score_sofar)
        running_score = torch.cat([running_score, logits[ar(Bn), 0, token][:, None]], dim=1)
        finished = finished * (token != ID_EOS).to(F32)
        if finished.sum() < 1:
            break
        token = token[:, None]
    return synth_sofar, score_sofar
