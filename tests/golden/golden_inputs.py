"""Seeded inputs shared by the golden-vector generator (which feeds them to the
reference) and the tests (which feed them to the build)."""
import re

import numpy as np

# name -> (width, height, kind, seed)
IMAGE_CASES = {
    "sq336": (336, 336, "noise", 0),        # BASELINE config 2: 1344x1344 HD, 17 crops, 2509 tokens
    "land640x480": (640, 480, "smooth", 1),  # 1008x1344 HD, 13 live crops
    "port500x1000": (500, 1000, "noise", 2),  # portrait -> transposed path
    "wide900x300": (900, 300, "smooth", 3),  # height needs white padding
}

TOKENIZE_TEXTS = ["ab", "abcde", "xyz"]
MERGE_PROMPT = "<|user|>\n<|image_1|>\nhi<|end|>\n<|assistant|>\n"


def make_image(w, h, kind, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    if kind == "noise":
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        ph = rng.uniform(0, 6.28, 3)
        a = np.stack([127.5 + 127.5 * np.sin(xx / (17.0 + 5 * c) + yy / (29.0 - 3 * c) + ph[c]) for c in range(3)], -1)
        a = a.clip(0, 255).astype(np.uint8)
    return Image.fromarray(a, "RGB")


# A mixed request set for the tiny vision model (serving / sharding tests): user prompts (request 0 carries an image) and
# the same prompts as the chat template renders them (phi_3_vision_mlx.py:346-351).  tests/golden/tiny_serve_oracle.npz
# holds the per-request B = 1 oracle tokens.
SERVE_PROMPTS = ["What is shown?", "hi", ("a longer question " * 6).strip(), "mid size prompt here", "x",
                 ("tell me more about it " * 3).strip(), "last one"]
SERVE_TEXTS = [f"<|user|>\n{'<|image_1|>' + chr(10) if i == 0 else ''}{p}<|end|>\n<|assistant|>\n" for i, p in enumerate(SERVE_PROMPTS)]
SERVE_STEPS = 6


def serve_requests(proc):
    """B = 1 model inputs of SERVE_TEXTS (request 0 carries the seeded 336x336 noise image)."""
    return [proc(t, [make_image(336, 336, "noise", 0)]) if "<|image_1|>" in t else proc(t) for t in SERVE_TEXTS]


class _Enc:
    def __init__(self, input_ids):
        self.input_ids = input_ids


class FakeTokenizer:
    """BOS(1)-prepending character tokenizer with single ids for chat markers."""
    SPECIAL = {"<|user|>": 32010, "<|end|>": 32007, "<|assistant|>": 32001}

    def _one(self, t):
        ids = [1]
        for part in re.split("(" + "|".join(re.escape(k) for k in self.SPECIAL) + ")", t):
            if part in self.SPECIAL:
                ids.append(self.SPECIAL[part])
            else:
                ids.extend(100 + ord(ch) for ch in part)
        return ids

    def __call__(self, texts):
        if isinstance(texts, str):
            return _Enc(self._one(texts))
        return _Enc([self._one(t) for t in texts])


# synthetic requests of the BASELINE configs live in the package (bench.py uses them too)
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from phi_3_vision_mlx_amd.workloads import c4_share, text_request, vqa_request  # noqa: E402,F401
