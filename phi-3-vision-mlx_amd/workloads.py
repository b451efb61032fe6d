"""Synthetic requests of the BASELINE configs -- shared by bench.py, the oracle fixture
generator and the GPU parity tests, so that what is benchmarked is what is pinned."""
import numpy as np


def vqa_request(img_processor, seed=0, dtype=np.float32, device=None):
    """BASELINE config 2: one seeded 336x336 uint8 noise image (-> 1344x1344 HD, 17 crops, 2509 image tokens) + 20 random
    text ids laid out as `_merge` would (BOS, text, image slots, BOS again (Q6), text).  seed 0 = bench.py's rank-0 request."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    img = Image.fromarray(rng.integers(0, 256, (336, 336, 3), dtype=np.uint8))
    # device: the image stage on the GPU (p3v_resample_u8 + p3v_hd_preprocess, same bits as the host path)
    image_inputs = img_processor.device_call([img], device) if device is not None else img_processor([img])
    n_img = image_inputs["num_img_tokens"][0]
    text_ids = rng.integers(3, 32000, 20)
    ids = np.concatenate([[1], text_ids[:8], -np.ones(n_img, dtype=np.int64), [1], text_ids[8:]])[None].astype(np.int64)
    pv = image_inputs["pixel_values"]
    return {"input_ids": ids, "pixel_values": pv if device is not None else np.asarray(pv, dtype=dtype),
            "image_sizes": np.asarray(image_inputs["image_sizes"]), "positions": np.argwhere(ids < 0)}


def text_request(seed, lo=16, hi=256):
    """A text-only request of BASELINE config 4: random ids, length ~U[lo, hi]."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(lo, hi + 1))
    return {"input_ids": np.concatenate([[1], rng.integers(3, 32000, n - 1)])[None].astype(np.int64)}


def c4_share(img_processor, rank=0, per_gpu=8, dtype=np.float32, device=None):
    """One GPU's share of BASELINE config 4 (64 mixed requests over 8 GPUs): per_gpu/2 single-image VQA requests
    (image seeds rank*per_gpu/2 ...) followed by per_gpu/2 text prompts -- B=1 model inputs, request order."""
    h = per_gpu // 2
    return [vqa_request(img_processor, rank * h + i, dtype, device) for i in range(h)] + [text_request(rank * h + i) for i in range(h)]
