"""Host-side processors: tokenisation/left-padding, image-tag merging and the
HD image transform.  Drop-in for the reference's `Phi3FProcessor`,
`Phi3VProcessor` and `Phi3VImageProcessor` (reference phi.py:228-372): same
call signatures, same dictionary keys, same values.

Outputs are NumPy arrays (the model object moves them to the GPU); with
``return_mx`` kept only for signature compatibility.

`Phi3VImageProcessor` reproduces the reference bit for bit, including the
degenerate two-tap "bicubic" global crop -- but the
reference's 4-deep Python loop over 336x336x3 pixels (phi.py:365-371, ~5.6 s
per image) is evaluated as four vectorised gathers with the identical
floating-point association, see `interpolate_336`.
"""
import os
import re

import numpy as np

IMAGE_TAG = r"<\|image_\d+\|>"


# ---------------------------------------------------------------------------
# Tokenizers
# ---------------------------------------------------------------------------
class _Enc:
    def __init__(self, input_ids):
        self.input_ids = input_ids

    def __getitem__(self, k):
        return getattr(self, k)


class ByteTokenizer:
    """Fallback tokenizer for synthetic model directories (no tokenizer files).

    Mimics the observable behaviour of the Llama-style Phi-3 tokenizer that the
    reference relies on: a BOS (id 1) is prepended to every encoded string
    (Q6), a dummy-prefix token precedes the text (so that
    ``encode(text, add_special_tokens=False)[1:]`` drops it exactly as
    reference phi_3_vision_mlx.py:538 expects), the chat markers are single
    ids (``<|end|>`` = 32007, ``<|assistant|>`` = 32001), pad id is 0 and text
    is byte-level (ids 3..258)."""
    BOS, PAD, PREFIX = 1, 0, 29871
    SPECIALS = {"<|endoftext|>": 32000, "<|assistant|>": 32001, "<|system|>": 32006,
                "<|end|>": 32007, "<|user|>": 32010}

    def __init__(self):
        self._inv = {v: k for k, v in self.SPECIALS.items()}
        self._split = re.compile("(" + "|".join(re.escape(k) for k in self.SPECIALS) + ")")

    def encode(self, text, add_special_tokens=True):
        ids = [self.BOS] if add_special_tokens else []
        ids.append(self.PREFIX)
        for part in self._split.split(text):
            if part in self.SPECIALS:
                ids.append(self.SPECIALS[part])
            else:
                ids.extend(3 + b for b in part.encode("utf-8"))
        return ids

    def __call__(self, texts, add_special_tokens=True):
        if isinstance(texts, str):
            return _Enc(self.encode(texts, add_special_tokens))
        return _Enc([self.encode(t, add_special_tokens) for t in texts])

    def decode(self, ids, skip_special_tokens=False):
        out, buf = [], bytearray()

        def flush():
            if buf:
                out.append(buf.decode("utf-8", errors="replace"))
                buf.clear()
        for i in (int(t) for t in ids):
            if 3 <= i <= 258:
                buf.append(i - 3)
                continue
            flush()
            if i == self.PREFIX or i in (self.BOS, self.PAD):
                continue
            out.append(self._inv[i] if i in self._inv else f"<{i}>")
        flush()
        return "".join(out)

    def batch_decode(self, seqs, skip_special_tokens=False):
        return [self.decode(s, skip_special_tokens) for s in seqs]


def load_tokenizer(local_dir):
    """`AutoTokenizer.from_pretrained(local_dir)` as the reference does
    (phi.py:230) when tokenizer files exist; otherwise the byte fallback."""
    has_files = local_dir is not None and any(
        os.path.exists(os.path.join(local_dir, f)) for f in ("tokenizer.json", "tokenizer.model", "tokenizer_config.json"))
    if has_files:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(local_dir)
    return ByteTokenizer()


# ---------------------------------------------------------------------------
# Text processor
# ---------------------------------------------------------------------------
class Phi3FProcessor:
    """reference phi.py:228-250."""

    def __init__(self, local_dir=None, return_mx=True, tokenizer=None):
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(local_dir)
        self.return_mx = return_mx

    def _tokenize(self, texts):
        if isinstance(texts, str):
            return {"input_ids": np.asarray(self.tokenizer(texts).input_ids, dtype=np.int64)[None]}
        ids = self.tokenizer(texts).input_ids
        width = max(len(s) for s in ids)
        # left pad: token 0, position id 1, mask 0 (phi.py:238-240)
        pids = [[1] * (width - len(s)) + list(range(len(s))) for s in ids]
        mask = [[0] * (width - len(s)) + [1] * len(s) for s in ids]
        ids = [[0] * (width - len(s)) + list(s) for s in ids]
        return {"input_ids": np.asarray(ids, dtype=np.int64), "pids": np.asarray(pids, dtype=np.int64),
                "mask": np.asarray(mask, dtype=np.int64)}

    def __call__(self, texts, images=None):
        if images is not None:
            print("WARNING: You are using phi3_mini_128k. Use phi3_v for VLM tasks.")
        return self._tokenize(texts)


class Phi3VProcessor(Phi3FProcessor):
    """reference phi.py:252-281."""

    def __init__(self, local_dir=None, return_mx=True, tokenizer=None):
        super().__init__(local_dir, return_mx, tokenizer)
        self.img_processor = Phi3VImageProcessor()

    def __call__(self, texts, images=None):
        if images is None:
            return self._tokenize(texts)
        if self.return_mx and os.environ.get("P3V_HOST_PREPROCESS") != "1":
            import torch
            if torch.cuda.is_available():                        # resize / pad / normalise / crop on the GPU, same bits
                return self._merge(self.img_processor.device_call(images, f"cuda:{torch.cuda.current_device()}"), texts)
        return self._merge(self.img_processor(images, dtype=np.float32 if self.return_mx else np.float64), texts)

    def _to_device(self, pixel_values):
        """The reference hands the model `mx.array(images)` (phi.py:279): float64 -> float32 ON the device, inside
        the processor, i.e. before the prefill timer starts.  With return_mx (default) we do the same: a float32
        torch tensor on the current GPU (CPU tensor when there is no GPU); return_mx=False keeps the float64 array."""
        if not self.return_mx:
            return pixel_values
        import torch
        if torch.is_tensor(pixel_values):                        # already produced on the device (device_call)
            return pixel_values
        t = torch.as_tensor(np.asarray(pixel_values), dtype=torch.float32)
        return t.cuda(non_blocking=True) if torch.cuda.is_available() else t

    def _merge(self, images, texts):
        # Each text chunk is tokenised on its own, so whatever the tokenizer
        # prepends (BOS) re-appears after the image slots (Q6) -- kept.
        chunks = self.tokenizer(re.split(IMAGE_TAG, texts)).input_ids
        n_tok = images["num_img_tokens"]
        tags = re.findall(IMAGE_TAG, texts)
        image_ids = [int(t.split("|")[1].split("_")[-1]) for t in tags]
        pads = [[-iid] * n_tok[iid - 1] for iid in image_ids]
        if len(chunks) > len(pads):
            pads = pads + [[]]
        input_ids = []
        for chunk, pad in zip(chunks, pads):
            input_ids.extend(chunk)
            input_ids.extend(pad)
        input_ids = np.asarray(input_ids, dtype=np.int64)[None]
        return {"input_ids": input_ids,
                "pixel_values": self._to_device(images["pixel_values"]),
                "image_sizes": np.asarray(images["image_sizes"], dtype=np.int64),
                "positions": np.argwhere(input_ids < 0)}


def collate_requests(requests, width=None):
    """B = 1 model inputs (`processor(text)` / `processor(text, images)` results) -> ONE left-padded batch.

    The reference batches text prompts only (`_tokenize`, phi.py:233-245) and runs image prompts at B = 1
    (phi_3_vision_mlx.py:377-378, `_merge` phi.py:276); BASELINE config 4 batches mixed image + text requests, so the
    batch is built with `_tokenize`'s own conventions -- left pad with id 0, position id 1 and mask 0 on the pad,
    positions 0..n-1 on the tokens -- and every image's slot positions move to (batch row, column + pad).  A row of the
    batch then sees exactly what its B = 1 run sees (pad keys get zero weight, Q7), which is what the parity tests check."""
    ids = [np.asarray(r["input_ids"]).reshape(-1) for r in requests]
    width = max(max(len(s) for s in ids), width or 0)            # `width`: the GLOBAL longest prompt when the batch is sharded
    out = {"input_ids": np.asarray([[0] * (width - len(s)) + s.tolist() for s in ids], dtype=np.int64),
           "pids": np.asarray([[1] * (width - len(s)) + list(range(len(s))) for s in ids], dtype=np.int64),
           "mask": np.asarray([[0] * (width - len(s)) + [1] * len(s) for s in ids], dtype=np.int64)}
    pix, sizes, pos = [], [], []
    for row, (r, s) in enumerate(zip(requests, ids)):
        if r.get("pixel_values") is None:
            continue
        pix.append(r["pixel_values"]), sizes.append(np.asarray(r["image_sizes"]))
        p = np.asarray(r["positions"]).copy()
        p[:, 0], p[:, 1] = row, p[:, 1] + (width - len(s))
        pos.append(p)
    if pix:
        import torch
        if any(torch.is_tensor(p) for p in pix):
            dev = next(p.device for p in pix if torch.is_tensor(p))
            out["pixel_values"] = torch.cat([p if torch.is_tensor(p) else torch.as_tensor(np.asarray(p), dtype=torch.float32).to(dev)
                                             for p in pix], dim=0)
        else:
            out["pixel_values"] = np.concatenate([np.asarray(p) for p in pix], axis=0)
        out["image_sizes"], out["positions"] = np.concatenate(sizes, axis=0), np.concatenate(pos, axis=0)
    return out


# ---------------------------------------------------------------------------
# Image processor
# ---------------------------------------------------------------------------
def _cubic(x):
    """Keys cubic (a=-0.5) on a NumPy float64 scalar -- scalar arithmetic on
    purpose: the reference evaluates it per element on scalars (phi.py:334-340)
    and scalar vs SIMD `pow` may differ in the last bit."""
    ax = np.abs(x)
    ax2 = ax ** 2
    ax3 = ax ** 3
    return ((1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1) +
            (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) & (ax <= 2)))


def _taps(scale, out_size, in_size):
    """Four-tap weights (fp32) and source indices per output coordinate
    (reference `get_weights_and_indices`, phi.py:333-359).  Only taps 0,1 are
    ever filled; taps 2,3 keep weight 0 and index 0, exactly as there."""
    oc = np.linspace(0, in_size - 1, out_size)
    ic = oc / scale
    left = np.floor(ic - 0.5).astype(np.int32)
    right = left + 1
    left = np.clip(left, 0, in_size - 1)
    right = np.clip(right, 0, in_size - 1)
    w = np.zeros((out_size, 4), dtype=np.float32)
    idx = np.zeros((out_size, 4), dtype=np.int32)
    idx[:, 0], idx[:, 1] = left, right
    for i in range(out_size):
        w[i, 0] = _cubic(ic[i] - left[i])
        w[i, 1] = _cubic(right[i] - ic[i])
        s = w[i].sum()
        if s != 0:
            w[i] /= s
    return w, idx


def pil_bilinear_coeffs(in_size, out_size):
    """Coefficients of Pillow's ImagingResample (libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc) for the
    triangle filter `Image.BILINEAR` selects, restated: per output coordinate the first input index, the tap count and the
    taps as 22-bit fixed point.  Same double-precision operations in the same order (the tap sum is accumulated tap by
    tap), so the integers -- and with them the resized image -- equal Pillow's (tests/test_processor_golden.py)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    ss = 1.0 / fs
    w = np.zeros((out_size, ksize))
    ww = np.zeros(out_size)
    for x in range(ksize):
        a = np.abs((x + xmin - center + 0.5) * ss)
        w[:, x] = np.where((x < xmax) & (a < 1.0), 1.0 - a, 0.0)
        ww = ww + w[:, x]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + w * (1 << 22)).astype(np.int32)
    return kk, np.stack([xmin, xmax], axis=1).astype(np.int32)


class Phi3VImageProcessor:
    """reference phi.py:283-372."""

    def __init__(self):
        self.num_crops = 16
        self.image_mean = np.array([0.48145466, 0.4578275, 0.40821073])
        self.image_std = np.array([0.26862954, 0.26130258, 0.27577711])

    def hd_transform(self, img):
        """`HD_transform` (phi.py:290-310): RGB -> (transpose if portrait) ->
        bilinear resize to (scale*336, .) -> white-pad height to a multiple of
        336 -> normalise -> CHW float64."""
        from PIL import Image, ImageOps
        img = img.convert("RGB")
        w, h = img.size
        portrait = w < h
        if portrait:
            img = img.transpose(Image.TRANSPOSE)
            w, h = img.size
        scale = int(np.sqrt(self.num_crops * w / h))
        img = img.resize([int(scale * 336), int(scale * 336 * h / w)], Image.BILINEAR)
        hh = img.size[1]
        diff = int(np.ceil(hh / 336) * 336) - hh
        top = int(diff / 2)
        img = ImageOps.expand(img, border=(0, top, 0, diff - top), fill=(255, 255, 255))
        if portrait:
            img = img.transpose(Image.TRANSPOSE)
        # (u8 / 255.0 - mean) / std (phi.py:309) through a 256-entry table per channel built with that very float64
        # expression: bit-identical, and 5.4 M divisions cheaper on a 1344 x 1344 image
        a = np.asarray(img)
        lut = (np.arange(256)[:, None] / 255.0 - self.image_mean) / self.image_std          # [256, 3] float64
        return np.stack([lut[:, c][a[:, :, c]] for c in range(3)], axis=0)

    @staticmethod
    def interpolate_336(x):
        """Reference `interpolate_336` (phi.py:331-372), vectorised over pixels.

        out[i,j] = np.sum(hw[i][:,None] * ww[j] * x[hi[i]][:, wi[j]]) over a 4x4
        window.  The weight product is rounded to fp32 before it meets the
        float64 pixels, and the 16 doubles e[4a+b] are added in the order NumPy's
        add-reduce uses on the contiguous 4x4: identity (+0.0), then the unrolled
        pairwise sum r[k] = e[k] + e[k+8]; ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)).
        Taps 2 and 3 have weight 0, so 12 of the 16 terms are signed zeros (the
        pixels are finite): adding them never changes a non-zero partial sum, and
        the leading +0.0 turns any -0.0 total into the reference's +0.0.  What is
        left is 0.0 + ((e0 + e1) + (e4 + e5)) -- bit-identical (the golden sha256 of
        `pixel_values` in tests/golden pins it) at a quarter of the work."""
        N, C, H, W = x.shape
        hw, hi = _taps(336 / H, 336, H)
        ww, wi = _taps(336 / W, 336, W)

        def term(a, b):
            wab = hw[:, a][:, None] * ww[:, b][None, :]              # fp32 x fp32 -> fp32
            return wab * x[:, :, hi[:, a], :][:, :, :, wi[:, b]]      # fp32 x fp64 -> fp64
        out = 0.0 + ((term(0, 0) + term(0, 1)) + (term(1, 0) + term(1, 1)))
        return out.astype(x.dtype, copy=False)

    def device_call(self, images, device):
        """`__call__` with everything after the RGB conversion on the GPU (p3v_resample_u8 x 2, p3v_hd_preprocess):
        pixel_values comes back as a float32 torch tensor on `device`, bit-identical to `__call__(..., dtype=float32)`."""
        import torch
        with torch.cuda.device(torch.device(device)):            # ops launch on the CURRENT device's stream
            return self._device_call(images, device)

    def _device_call(self, images, device):
        import torch
        from . import ops
        out = torch.empty((len(images), 17, 3, 336, 336), dtype=torch.float32, device=device)
        lut = torch.as_tensor(np.ascontiguousarray(((np.arange(256)[:, None] / 255.0 - self.image_mean) / self.image_std).T)).to(device)
        shapes = []

        def dev(a):
            return torch.from_numpy(np.array(a, copy=True, order="C")).to(device)   # (np.asarray(PIL image) is read-only)
        for n, img in enumerate(images):
            a = np.asarray(img.convert("RGB"))
            h, w = a.shape[:2]
            portrait = w < h
            if portrait:                                         # Image.TRANSPOSE (phi.py:295-297)
                a = a.transpose(1, 0, 2)
                h, w = a.shape[:2]
            scale = int(np.sqrt(self.num_crops * w / h))
            rw, rh = int(scale * 336), int(scale * 336 * h / w)
            hp = int(np.ceil(rh / 336) * 336)
            top = int((hp - rh) / 2)
            t = dev(a)
            if rw != w:
                t = ops.resample_u8(t, rw, 1, *map(dev, pil_bilinear_coeffs(w, rw)))
            if rh != h:
                t = ops.resample_u8(t, rh, 0, *map(dev, pil_bilinear_coeffs(h, rh)))
            H, W = (rw, hp) if portrait else (hp, rw)
            hw, hi = _taps(336 / H, 336, H)
            ww, wi = _taps(336 / W, 336, W)
            ops.hd_preprocess(t, top, hp, portrait, lut, dev(hw[:, :2]), dev(hi[:, :2]), dev(ww[:, :2]), dev(wi[:, :2]), out[n])
            shapes.append([H, W])
        num_img_tokens = [int((h // 336 * w // 336 + 1) * 144 + 1 + (h // 336 + 1) * 12) for h, w in shapes]
        return {"pixel_values": out, "image_sizes": shapes, "num_img_tokens": num_img_tokens}

    def __call__(self, images, dtype=np.float64):
        """-> pixel_values [n, 17, 3, 336, 336] `dtype` (float64 like the reference; float32 = the SAME values after the
        cast the reference applies with mx.array(...) (phi.py:279), rounded once either way), image_sizes, num_img_tokens."""
        hd = [self.hd_transform(im) for im in images]
        shapes = [[im.shape[1], im.shape[2]] for im in hd]
        num_img_tokens = [int((h // 336 * w // 336 + 1) * 144 + 1 + (h // 336 + 1) * 12) for h, w in shapes]
        out = np.zeros((len(hd), 17, 3, 336, 336), dtype=dtype)  # 17 crop slots, zero-padded (phi.py:311-316)
        for i, (im, (h, w)) in enumerate(zip(hd, shapes)):
            hc, wc = h // 336, w // 336
            out[i, 0] = self.interpolate_336(im[None])[0]        # global view first (phi.py:312)
            # sub-crops in (row, column) order, written straight into their slots: one strided copy
            out[i, 1:1 + hc * wc].reshape(hc, wc, 3, 336, 336)[...] = im.reshape(3, hc, 336, wc, 336).transpose(1, 3, 0, 2, 4)
        return {"pixel_values": out, "image_sizes": shapes, "num_img_tokens": num_img_tokens}
