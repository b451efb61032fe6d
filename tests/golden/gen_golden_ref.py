"""Generate golden vectors from the REFERENCE's own host code.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the committed outputs (`ref_processor.npz`,
`ref_processor.json`).  The reference's `phi.py` imports `mlx`, which is not
installable here, so the import is satisfied with inert stub modules -- only
the NumPy/PIL classes (`Phi3VImageProcessor`, `Phi3FProcessor._tokenize`,
`Phi3VProcessor._merge`) are executed, and those never touch MLX arithmetic
(`mx.array` is bound to `np.asarray`).

    python tests/golden/gen_golden_ref.py
"""
import hashlib
import importlib.machinery
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from golden_inputs import IMAGE_CASES, MERGE_PROMPT, TOKENIZE_TEXTS, FakeTokenizer, make_image  # noqa: E402


def _stub_mlx():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any

    def mk(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)

        def _ga(k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any
        m.__getattr__ = _ga
        sys.modules[name] = m
        return m
    mlx, core, nn, utils, optim = (mk(n) for n in ("mlx", "mlx.core", "mlx.nn", "mlx.utils", "mlx.optimizers"))
    core.compile = lambda f=None, **k: f if f is not None else (lambda g: g)
    core.array = np.asarray
    nn.Module = type("Module", (), {})
    mlx.core, mlx.nn, mlx.utils, mlx.optimizers = core, nn, utils, optim


def main():
    _stub_mlx()
    sys.path.insert(0, "/root/reference")
    import phi  # the reference module

    ip = phi.Phi3VImageProcessor()
    arrays, meta = {}, {"images": []}
    for name, (w, h, kind, seed) in IMAGE_CASES.items():
        img = make_image(w, h, kind, seed)
        out = ip([img])
        pv = np.asarray(out["pixel_values"])
        assert pv.dtype == np.float64 and pv.shape[1:] == (17, 3, 336, 336)
        arrays[f"{name}_pv_sub"] = pv[0, :, :, ::24, ::24].copy()           # 17x3x14x14 float64 sample
        arrays[f"{name}_pv_glb_corner"] = pv[0, 0, :, :96, :96].copy()       # global crop, live corner
        meta["images"].append({
            "name": name, "w": w, "h": h, "kind": kind, "seed": seed,
            "image_sizes": [list(map(int, s)) for s in out["image_sizes"]],
            "num_img_tokens": [int(t) for t in out["num_img_tokens"]],
            "pixel_values_sha256": hashlib.sha256(np.ascontiguousarray(pv).tobytes()).hexdigest(),
            "pixel_values_sum": float(pv.sum()),
            "global_nonzero": int(np.count_nonzero(pv[0, 0])),
        })
        print(name, meta["images"][-1])

    # tokenise / merge with a fake BOS-prepending tokenizer
    fp = phi.Phi3VProcessor.__new__(phi.Phi3VProcessor)
    fp.tokenizer, fp.return_mx, fp.img_processor = FakeTokenizer(), True, ip
    tok = fp._tokenize(TOKENIZE_TEXTS)
    meta["tokenize"] = {k: np.asarray(v).tolist() for k, v in tok.items()}
    meta["tokenize_str"] = {k: np.asarray(v).tolist() for k, v in fp._tokenize(TOKENIZE_TEXTS[0]).items()}
    w, h, kind, seed = IMAGE_CASES["sq336"]
    merged = fp(MERGE_PROMPT, [make_image(w, h, kind, seed)])
    ids = np.asarray(merged["input_ids"])
    pos = np.asarray(merged["positions"])
    meta["merge"] = {
        "keys": sorted(merged.keys()), "input_ids_shape": list(ids.shape),
        "n_negative": int((ids < 0).sum()), "first_neg": int(np.argmax(ids[0] < 0)),
        "head": ids[0, :int(np.argmax(ids[0] < 0))].tolist(),
        "tail": ids[0, int(np.argmax(ids[0] < 0)) + int((ids < 0).sum()):].tolist(),
        "positions_shape": list(pos.shape), "positions_first": pos[0].tolist(), "positions_last": pos[-1].tolist(),
        "image_sizes": np.asarray(merged["image_sizes"]).tolist(),
    }
    np.savez_compressed(os.path.join(HERE, "ref_processor.npz"), **arrays)
    with open(os.path.join(HERE, "ref_processor.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote ref_processor.{npz,json}")


if __name__ == "__main__":
    main()
