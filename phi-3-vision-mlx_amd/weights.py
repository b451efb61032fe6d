"""Weight tables for the hot path: names/shapes in the HF layout the reference
loads (reference phi_3_vision_mlx.py:371-374), a seeded
synthetic generator that is bit-identical on CPU and GPU, and a safetensors
directory loader.

The synthetic generator is counter-based (a 32-bit integer hash of the element
index, four hashed bytes summed -> Irwin-Hall(4) ~ normal), written with exact
integer/fp32 torch ops only so the same values come out on any device.  It is
data generation for tests/bench, not part of the compute path.
"""
import glob
import os
import zlib

import torch

from .config import is_vision

_M32 = 0xFFFFFFFF


def _hash32(x):
    """lowbias32 on int64 tensors holding uint32 values."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _byte_sums(s, e, seed, device):
    """Irwin-Hall(4) integers in [0, 1020] (mean 510, sd 147.8) for element indices [s, e)."""
    base = (int(seed) * 0x9E3779B1) & _M32
    idx = torch.arange(s, e, dtype=torch.int64, device=device)
    h = _hash32((idx + base) & _M32)
    return (h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF)


def synth_values(n, seed, std, mean=0.0, device="cpu", dtype=torch.bfloat16, chunk=None):
    """n pseudo-normal values with the given std/mean; deterministic in (seed, index)."""
    if chunk is None:      # CPU: stay cache-resident (8 MB temporaries); GPU: amortise launches
        chunk = (1 << 20) if str(device) == "cpu" else (1 << 24)
    out = torch.empty(n, dtype=dtype, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        b = _byte_sums(s, e, seed, device)
        v = (b.to(torch.float32) - 510.0) * (float(std) / 147.80054127) + float(mean)
        out[s:e] = v.to(dtype)
    return out


def row_exponents(n_rows, seed, spread, device="cpu"):
    """Integer exponents e_v ~ round(spread * N(0,1)) - const for a heavy-tailed (log2-normal) per-row scale 2**e_v:
    integer arithmetic only, so CPU and GPU agree bit for bit.  The constant keeps the largest rows near 2**2."""
    b = _byte_sums(0, n_rows, seed, device) - 510                         # ~ 147.8 * z
    num = int(round(float(spread) * 1024))
    e = torch.div(b * num + 1024 * 74, 1024 * 148, rounding_mode="floor")
    return (e - int(round(3.0 * float(spread))) + 2).to(torch.int32)


def weight_specs(cfg):
    """[(name, shape, kind)] in HF naming; kind in {matrix, norm, bias, embed, gn}."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    hd = H // nh
    specs = [("model.embed_tokens.weight", (V, H), "embed")]
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        specs += [
            (p + "input_layernorm.weight", (H,), "norm"),
            (p + "self_attn.qkv_proj.weight", ((nh + 2 * nkv) * hd, H), "matrix"),
            (p + "self_attn.o_proj.weight", (H, nh * hd), "matrix"),
            (p + "post_attention_layernorm.weight", (H,), "norm"),
            (p + "mlp.gate_up_proj.weight", (2 * I, H), "matrix"),
            (p + "mlp.down_proj.weight", (H, I), "matrix"),
        ]
    specs += [("model.norm.weight", (H,), "norm"), ("lm_head.weight", (V, H), "matrix")]
    if is_vision(cfg):
        c = cfg.clip
        D, DI, P, C = c["hidden_size"], c["intermediate_size"], c["patch_size"], c["num_channels"]
        npos = (c["image_size"] // P) ** 2 + 1
        v = "model.vision_embed_tokens.img_processor.vision_model."
        specs += [
            (v + "embeddings.class_embedding", (D,), "bias"),
            (v + "embeddings.patch_embedding.weight", (D, C, P, P), "matrix"),
            (v + "embeddings.position_embedding.weight", (npos, D), "matrix"),
            (v + "pre_layrnorm.weight", (D,), "norm"),
            (v + "pre_layrnorm.bias", (D,), "bias"),
        ]
        for j in range(c["num_hidden_layers"]):
            q = v + f"encoder.layers.{j}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                specs += [(q + f"self_attn.{nm}.weight", (D, D), "matrix"),
                          (q + f"self_attn.{nm}.bias", (D,), "bias")]
            specs += [
                (q + "layer_norm1.weight", (D,), "norm"), (q + "layer_norm1.bias", (D,), "bias"),
                (q + "mlp.fc1.weight", (DI, D), "matrix"), (q + "mlp.fc1.bias", (DI,), "bias"),
                (q + "mlp.fc2.weight", (D, DI), "matrix"), (q + "mlp.fc2.bias", (D,), "bias"),
                (q + "layer_norm2.weight", (D,), "norm"), (q + "layer_norm2.bias", (D,), "bias"),
            ]
        specs += [(v + "post_layernorm.weight", (D,), "norm"), (v + "post_layernorm.bias", (D,), "bias")]
        e = "model.vision_embed_tokens."
        Do = cfg.img_processor["image_dim_out"]
        specs += [
            (e + "glb_GN", (1, 1, 4 * Do), "gn"), (e + "sub_GN", (1, 1, 1, 4 * Do), "gn"),
            (e + "img_projection.0.weight", (H, 4 * Do), "matrix"), (e + "img_projection.0.bias", (H,), "bias"),
            (e + "img_projection.2.weight", (H, H), "matrix"), (e + "img_projection.2.bias", (H,), "bias"),
        ]
    return specs


_KIND = {  # kind -> (std, mean)
    "matrix": (0.02, 0.0), "embed": (0.02, 0.0), "bias": (0.02, 0.0),
    "gn": (0.02, 0.0), "norm": (0.05, 1.0),
}


def synth_tensor(name, shape, kind, seed=0, device="cpu", std_scale=1.0):
    n = 1
    for s in shape:
        n *= s
    std, mean = _KIND[kind]
    if kind in ("matrix", "embed"):
        std = std * std_scale
    sd = (zlib.crc32(name.encode()) ^ (seed * 0x85EBCA6B)) & _M32
    return synth_values(n, sd, std, mean, device=device).reshape(shape)


def peaked_lm_head(w, spread, head_seed=0):
    """lm_head rows times seeded power-of-two factors 2**e_v, e_v ~ round(spread * N(0,1)) (exact in bf16).

    Random N(0, s) rows give Gaussian logits whose top-2 gap is ~5 % of max|logit| -- a few bf16 ulps of the logits
    themselves, so a greedy argmax is decided by rounding noise and token parity cannot be asserted.  Trained models
    are decisive; a heavy-tailed row norm reproduces that (median top-2 gap ~20 % of max|logit| at spread 4) while the
    winner still depends on the whole hidden state (~10-20 effective candidates).  Parity fixtures only; bench.py
    keeps the plain N(0, 0.02) head (timing does not depend on the values)."""
    e = row_exponents(w.shape[0], 0xC0FFEE ^ (int(head_seed) * 0x27D4EB2F), spread, device=w.device)
    return torch.ldexp(w.float(), e[:, None]).to(w.dtype)


OUTLIERS = dict(n=6, gain=64.0, kv_n=2, kv_gain=8.0)     # `outliers=True`


def outlier_channels(cfg, n, seed=0):
    """The seeded hidden channels that carry the heavy tail (sorted)."""
    g = torch.Generator().manual_seed(0x0DD1E5 ^ (int(seed) * 0x9E3779B1 & _M32))
    return torch.randperm(cfg.hidden_size, generator=g)[:n].sort().values


def add_outliers(cfg, w, seed=0, n=6, gain=64.0, kv_n=2, kv_gain=8.0):
    """Heavy-tailed activations, as trained decoders show them (a handful of residual-stream channels 50-100x the rest,
    large key / value dimensions): N(0, s) weights give Gaussian activations, under which a per-row e4m3 activation scale
    (`p3v_quant_fp8_rows`, phi_3_vision_mlx.py:291-305's quantised path), the int8 KV scales and the bf16 roundings of the
    attention are never stressed.  In place, all factors powers of two (exact in bf16):
      * `n` seeded hidden channels x `gain` wherever the residual stream is WRITTEN: embedding columns, o_proj / down_proj
        output rows of every layer, the image projector's output rows (+ bias) -- so text and image tokens carry them from
        layer 0 on and every layer keeps feeding them.  RMSNorm then spends most of its range on those channels: with
        H = 3072, n = 6, gain = 64 the other channels of a normalised row shrink ~3x, and a per-row fp8 scale (row max / 448)
        leaves them ~3 bits;
      * `kv_n` dimensions per key and value head x `kv_gain` (qkv_proj output rows): per-token int8 KV scales are set by
        those dimensions."""
    ch = outlier_channels(cfg, n, seed).to(next(iter(w.values())).device)
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    hd = cfg.hidden_size // nh
    w["model.embed_tokens.weight"][:, ch] *= gain
    g = torch.Generator().manual_seed(0xCAFE ^ seed)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "self_attn.o_proj.weight"][ch, :] *= gain
        w[p + "mlp.down_proj.weight"][ch, :] *= gain
        if kv_n:
            dims = torch.randperm(hd, generator=g)[:kv_n]
            rows = torch.cat([(nh + h) * hd + dims for h in range(2 * nkv)]).to(ch.device)     # K heads, then V heads
            w[p + "self_attn.qkv_proj.weight"][rows, :] *= kv_gain
    pj = "model.vision_embed_tokens.img_projection.2."
    if pj + "weight" in w:
        w[pj + "weight"][ch, :] *= gain
        w[pj + "bias"][ch] *= gain
    return w


def residual_scale_default(cfg):
    """1 / sqrt(2 * n_layers): the scale of the residual-branch output projections under which a random decoder stops amplifying."""
    return (2.0 * cfg.num_hidden_layers) ** -0.5


def synth_weights(cfg, seed=0, device="cpu", std_scale=1.0, lm_head_spread=0.0, lm_head_seed=0, outliers=None, residual_scale=None):
    """Seeded synthetic bf16 weights for every tensor of `weight_specs(cfg)`.

    std_scale > 1 sharpens the logits of tiny test models (wider top-2 margins);
    lm_head_spread > 0 makes the greedy argmax decisive (`peaked_lm_head`);
    outliers: None | True (= OUTLIERS) | dict of `add_outliers` arguments -> heavy-tailed activations;
    residual_scale: the WELL-CONDITIONED checkpoint of the parity tests (round 5) -- the decoder's residual-branch output
    projections (o_proj, down_proj) times this factor (True = 1 / sqrt(2 * n_layers), the usual depth-scaled initialisation).
    With plain N(0, 0.02) everywhere each of the 32 layers adds a branch as large as the stream itself and a single bf16
    rounding grows to ~6 % of the logits; depth-scaled, the stream is dominated by the embedding and the
    error of two correct implementations stays near 1 %."""
    w = {n: synth_tensor(n, s, k, seed, device, std_scale) for n, s, k in weight_specs(cfg)}
    if residual_scale:
        f = residual_scale_default(cfg) if residual_scale is True else float(residual_scale)
        for n in w:
            if n.startswith("model.layers.") and (n.endswith("self_attn.o_proj.weight") or n.endswith("mlp.down_proj.weight")):
                w[n] = (w[n].float() * f).to(w[n].dtype)
    if outliers:
        add_outliers(cfg, w, seed, **(OUTLIERS if outliers is True else outliers))
    if lm_head_spread:
        w["lm_head.weight"] = peaked_lm_head(w["lm_head.weight"], lm_head_spread, lm_head_seed)
    return w


def load_safetensors_dir(model_path, cfg, device="cpu"):
    """Read every ``*.safetensors`` under `model_path` (reference `_get_wt`,
    phi_3_vision_mlx.py:371-374).  The conv weight stays in HF ``[O,C,kh,kw]``
    order (the patch-embed kernel unfolds patches in the matching (c,ky,kx)
    order); a 'sanitized' MLX re-save (``[O,kh,kw,C]``) is permuted back."""
    from safetensors import safe_open
    out, raw = {}, {}
    files = sorted(glob.glob(f"{model_path}/*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {model_path}")
    for wf in files:
        with safe_open(wf, framework="pt", device="cpu") as f:
            for k in f.keys():
                raw[k] = f.get_tensor(k)
    quantized = getattr(cfg, "quantized", None) or None
    for k, t in raw.items():
        if quantized and (k.endswith(".scales") or k.endswith(".biases")):
            continue
        base = k[:-len(".weight")] if k.endswith(".weight") else None
        if quantized and base is not None and base + ".scales" in raw:
            # MLX nn.quantize checkpoint (quantized_model.safetensors, phi_3_vision_mlx.py:297-305): uint32 codes + per-group
            # scales / biases.  Decoder projections and lm_head keep their 4-bit form (Q4Weight -> p3v_gemv_q4); embeddings,
            # the ViT and the projector are dequantised here (prefill-only / tiny), exactly scale * q + bias.
            if int(quantized.get("bits", 4)) != Q4_BITS or int(quantized.get("group_size", 64)) != Q4_GROUP:
                raise NotImplementedError(f"only {Q4_BITS}-bit group-{Q4_GROUP} MLX checkpoints are supported, got {quantized}")
            q = Q4Weight(t.view(torch.int32) if t.dtype != torch.int32 else t, raw[base + ".scales"], raw[base + ".biases"])
            fast = k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith("_proj.weight"))
            out[k] = q if fast else mlx_dequantize(*q, dtype=torch.bfloat16).to(device)
            continue
        if "patch_embedding.weight" in k and getattr(cfg, "sanitized", False):
            t = t.permute(0, 3, 1, 2).contiguous()
        out[k] = t.to(torch.bfloat16).to(device)
    missing = [n for n, _, _ in weight_specs(cfg) if n not in out]
    if missing:
        raise KeyError(f"weights missing from {model_path}: {missing[:4]}{'...' if len(missing) > 4 else ''}")
    return out


def save_safetensors_dir(weights, cfg_dict, model_path):
    """Write a model dir (config.json + model.safetensors) in the HF layout."""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(model_path, exist_ok=True)
    save_file({k: v.detach().cpu().contiguous() for k, v in weights.items()}, f"{model_path}/model.safetensors")
    with open(f"{model_path}/config.json", "w") as f:
        json.dump(cfg_dict, f, indent=1)


# ---------------------------------------------------------------- LoRA adapters (phi_3_vision_mlx.py:234-245, 266-271)
ADAPTER_CONFIG, ADAPTER_WEIGHTS = "adapter_config.json", "adapters.safetensors"


def save_adapter(path, lora_cfg, tensors):
    """Write `adapter_config.json` + `adapters.safetensors` in the reference's format (phi.py:56,61):
    lora_cfg = {model_path, adapter_path, lora_layers (int = last n | list of indices), lora_targets (module names
    inside a decoder layer, e.g. "self_attn.qkv_proj"), lora_parameters {rank, alpha, dropout, scale}};
    tensors = {"model.layers.<i>.<target>.lora_a": [in, r] f32, "...lora_b": [r, out] f32}."""
    import json
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, ADAPTER_CONFIG), "w") as f:
        json.dump(lora_cfg, f, indent=4)
    save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, os.path.join(path, ADAPTER_WEIGHTS))


def load_adapter(path):
    """-> (lora_cfg dict, {name: tensor}) from an adapter directory; FileNotFoundError like the reference's
    `_get_cfg` / `load_weights` when either file is missing."""
    import json
    from safetensors.torch import load_file
    cfg_file, wt_file = os.path.join(path, ADAPTER_CONFIG), os.path.join(path, ADAPTER_WEIGHTS)
    for f in (cfg_file, wt_file):
        if not os.path.exists(f):
            raise FileNotFoundError(f)
    with open(cfg_file) as f:
        lora_cfg = json.load(f)
    return lora_cfg, load_file(wt_file)


def resolve_adapter(cfg, lora_cfg, tensors, device="cpu"):
    """`_linear_to_lora_layers` (phi_3_vision_mlx.py:234-245) + `load_weights(strict=False)` (:271):
    -> {"model.layers.<i>.<target>.weight": (lora_a [in,r] f32, lora_b [r,out] f32, scale)} for every adapted projection.
    lora_layers int n = the LAST n decoder layers, list = layer indices; scale = cfg.scale * alpha / rank (phi.py:120).
    A LoRA layer whose tensors are absent from the file keeps its initial lora_b = 0, i.e. is an identity: skipped."""
    n_layers = cfg.num_hidden_layers
    layers = lora_cfg["lora_layers"]
    if isinstance(layers, int):
        idx = list(range(n_layers))[-layers:] if layers > 0 else []
    elif isinstance(layers, list):
        idx = [i % n_layers for i in layers]
    else:
        raise ValueError("Invalid type for lora_layers. Expected int (number of layers) or list (layer indices or names).")
    prm = lora_cfg["lora_parameters"]
    scale = float(prm["scale"]) * (float(prm["alpha"]) / float(prm["rank"]))
    out = {}
    for i in idx:
        for tgt in lora_cfg["lora_targets"]:
            a, b = tensors.get(f"model.layers.{i}.{tgt}.lora_a"), tensors.get(f"model.layers.{i}.{tgt}.lora_b")
            if a is None or b is None:
                continue
            if a.shape[1] != b.shape[0] or a.shape[1] > 64:
                raise ValueError(f"unsupported LoRA shapes for layer {i} {tgt}: {tuple(a.shape)} x {tuple(b.shape)} (rank <= 64)")
            out[f"model.layers.{i}.{tgt}.weight"] = (a.to(device, torch.float32).contiguous(), b.to(device, torch.float32).contiguous(), scale)
    return out


# ---------------------------------------------------------------- 4-bit group-64 weights (MLX nn.quantize format)
Q4_GROUP, Q4_BITS = 64, 4


class Q4Weight(tuple):
    """(packed int32 [N, K/8] in MLX's nibble order, scales [N, K/64], biases [N, K/64]) of one quantised projection."""
    def __new__(cls, packed, scales, biases):
        return super().__new__(cls, (packed, scales, biases))


def mlx_quantize(w, group_size=Q4_GROUP, bits=Q4_BITS):
    """Affine group quantisation as mlx 0.15.0's mx.quantize does it (the reference calls nn.quantize(model, 64, 4),
    phi_3_vision_mlx.py:264,297-305, and mx.quantize(keys, group_size=32), phi.py:531): per group of `group_size` input columns
    w ~ scale * q + bias,  q in 0..2^bits-1, the end of the range with the larger magnitude represented exactly.
    In that release the function is a composite of array primitives, so EVERY intermediate is an array of w's dtype: for bf16
    weights the range, the scale, edge / scale, edge / q0, w - bias and (w - bias) / scale each round to bf16 (`r` below); for
    fp32 inputs `r` is the identity.  (Later releases compute the group statistics in fp32 inside one kernel; a checkpoint
    written by either only needs `mlx_dequantize` below.)
    -> (packed uint32-as-int32 [N, K*bits/32] in MLX's order: weight k of a word at bits [4k, 4k+4); scales, biases
    [N, K/group] in w.dtype)."""
    N, K = w.shape
    dt = w.dtype

    def r(t):                                                  # one primitive's result: computed in fp32, stored in w's dtype
        return t.to(dt).float()
    g = w.float().reshape(N, K // group_size, group_size)
    n_bins = float((1 << bits) - 1)
    w_max, w_min = g.amax(-1), g.amin(-1)
    mask = w_min.abs() > w_max.abs()
    eps = r(torch.tensor(1e-7, dtype=torch.float32, device=w.device))
    scales = torch.maximum(r(r(w_max - w_min) / n_bins), eps)
    scales = torch.where(mask, scales, -scales)
    edge = torch.where(mask, w_min, w_max)
    q0 = torch.round(r(edge / scales))
    scales = torch.where(q0 != 0, r(edge / q0), scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    q = torch.round(r(r(g - biases[..., None]) / scales[..., None])).clamp(0, n_bins).to(torch.int64).reshape(N, K)
    per = 32 // bits
    shifts = (torch.arange(per, dtype=torch.int64, device=w.device) * bits)
    packed = (q.reshape(N, K // per, per) << shifts).sum(-1)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)     # uint32 bit pattern in int32
    return packed, scales.to(dt), biases.to(dt)


def mlx_unpack(packed, bits=Q4_BITS):
    """uint32 words (as int32 / uint32 tensor) -> integer codes [N, K]."""
    per = 32 // bits
    u = packed.to(torch.int64) & 0xFFFFFFFF
    shifts = (torch.arange(per, dtype=torch.int64, device=packed.device) * bits)
    return ((u[..., None] >> shifts) & ((1 << bits) - 1)).reshape(packed.shape[0], -1)


def mlx_dequantize(packed, scales, biases, group_size=Q4_GROUP, bits=Q4_BITS, dtype=None):
    """scale * q + bias per group.  dtype=None: in float32 (what a fused quantised matmul computes on the fly: p3v_gemv_q4,
    mx.quantized_matmul).  dtype given: as mlx 0.15.0's mx.dequantize ARRAY -- multiply then add, each rounding to `dtype` -- which
    is what nn.QuantizedEmbedding returns for a looked-up row (the reference's 4-bit embedding table, phi_3_vision_mlx.py:297-305)."""
    q = mlx_unpack(packed, bits).float()
    N, K = q.shape
    prod = q.reshape(N, K // group_size, group_size) * scales.float()[..., None]
    if dtype is not None:
        prod = prod.to(dtype).float()
    out = (prod + biases.float()[..., None]).reshape(N, K)
    return out if dtype is None else out.to(dtype)


def q4_repack(packed, scales, biases):
    """MLX layout -> the device layout of p3v_gemv_q4 (include/p3v.h): nibbles of weights 8d..8d+7 at bits
    0,16,4,20,8,24,12,28 of dword d; scale | bias << 16 as one dword per group.
    The device format holds scale and bias as bf16 -- the dtype the reference's `*_Q` checkpoints have (they are quantised
    from the bf16 HF checkpoint, phi_3_vision_mlx.py:291-305).  A checkpoint converted to fp16 / fp32 loses mantissa
    bits here and no longer dequantises to exactly scale * q + bias: that is reported, not hidden."""
    if scales.dtype != torch.bfloat16:
        lossy = bool((scales.to(torch.bfloat16).to(scales.dtype) != scales).any() or
                     (biases.to(torch.bfloat16).to(biases.dtype) != biases).any())
        if lossy:
            import warnings
            warnings.warn(f"q4_repack: {scales.dtype} scales / biases rounded to bf16 (device format); dequantised weights "
                          "differ from the checkpoint's by up to 2^-9 relative", stacklevel=2)
    q = mlx_unpack(packed).reshape(packed.shape[0], -1, 8)
    pos = torch.tensor([0, 16, 4, 20, 8, 24, 12, 28], dtype=torch.int64, device=packed.device)
    w4 = (q << pos).sum(-1)
    w4 = torch.where(w4 >= 2 ** 31, w4 - 2 ** 32, w4).to(torch.int32)
    s16 = scales.to(torch.bfloat16).view(torch.int16).to(torch.int64) & 0xFFFF
    b16 = biases.to(torch.bfloat16).view(torch.int16).to(torch.int64) & 0xFFFF
    sb = s16 | (b16 << 16)
    sb = torch.where(sb >= 2 ** 31, sb - 2 ** 32, sb).to(torch.int32)
    return w4.contiguous(), sb.contiguous()
