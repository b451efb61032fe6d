"""Import the REFERENCE's own Python (`/root/reference/phi.py`, `phi_3_vision_mlx.py`) over the functional MLX stand-in
(`mlx_shim.py`).  Build-container only: the reference's files never travel (nothing here copies their text anywhere; the
modules are executed from where they lie), and nothing under `-m gpu`, `smoke()` or `bench.py` imports this file.

`phi_3_vision_mlx.py:730` (inside `rag()`, out of scope) uses a Python-3.12 f-string that the image's Python 3.10 cannot
parse.  The loader compiles the file with that ONE statement replaced, in memory, by a `raise NotImplementedError`; every
other line -- `_load`, `_get_cfg`, `_get_wt`, `_generate`, `_choose_from`, `_constrain`, `Streamer`, the stoppers -- runs as
written.  Third-party modules the hot path never touches (`gradio`, the reference's `api`/`gte` helpers, which pull remote
services and a second model) are inert stubs.
"""
import ast
import importlib.machinery
import os
import re
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
REFERENCE = os.environ.get("P3V_REFERENCE", "/root/reference")
_CACHE = {}


def available():
    return os.path.exists(os.path.join(REFERENCE, "phi.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)


def load_reference():
    """-> (mx, phi, loops): the shim's mlx.core, the reference's `phi` module and its `phi_3_vision_mlx` module."""
    if _CACHE:
        return _CACHE["mx"], _CACHE["phi"], _CACHE["loops"]
    import mlx_shim
    mx, _ = mlx_shim.install()
    _stub("gradio")
    _stub("api", bark_api=None, mistral_api=None)
    _stub("gte", VDB=None, GteModel=None)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import phi
    path = os.path.join(REFERENCE, "phi_3_vision_mlx.py")
    lines = open(path).read().split("\n")
    patched = []
    for _ in range(8):
        try:
            tree = ast.parse("\n".join(lines), filename=path)
            break
        except SyntaxError as e:
            ind = re.match(r"\s*", lines[e.lineno - 1]).group(0)
            lines[e.lineno - 1] = ind + "raise NotImplementedError('python-3.12-only statement, off the pinned path')"
            patched.append(e.lineno)
    assert patched in ([], [730]), f"unexpected statements neutralised: {patched}"
    loops = types.ModuleType("phi_3_vision_mlx")
    loops.__file__ = path
    sys.modules["phi_3_vision_mlx"] = loops
    exec(compile(tree, path, "exec"), loops.__dict__)
    _CACHE.update(mx=mx, phi=phi, loops=loops)
    return mx, phi, loops


def load_model(model_dir, tokenizer, clip_cfg=None, adapter_path=None, **kwargs):
    """The reference's own `_load` (phi_3_vision_mlx.py:257-274) on an HF-layout directory.  Two injections, both data:
    `AutoTokenizer.from_pretrained` returns `tokenizer` (there are no tokenizer files), and the CLIP geometry the reference
    hard-codes as a class attribute (phi.py:375-384) is overridden with `clip_cfg` for tiny models."""
    mx, phi, loops = load_reference()

    class _Auto:
        @staticmethod
        def from_pretrained(local_dir):
            return tokenizer
    phi.AutoTokenizer = _Auto
    if clip_cfg is not None:
        phi.Phi3ImageEmbedding.CLIP_VIT_LARGE_PATCH14_336_CONFIG = types.SimpleNamespace(**clip_cfg)
    return loops._load(model_path=model_dir, adapter_path=adapter_path, **kwargs)


class Recorder:
    """Wraps the reference model: forwards every call, keeps (input_ids, kwargs, logits) per call."""

    def __init__(self, model):
        self.model, self.calls = model, []

    def __call__(self, *a, **k):
        logits, cache = self.model(*a, **k)
        ids = k.get("input_ids", a[0] if a else None)
        self.calls.append(dict(input_ids=ids, max_tokens=k.get("max_tokens", 0), advance_offset=k.get("advance_offset"),
                               n_beam=k.get("n_beam", 1), logits=logits, offset=(cache[0].offset if cache else None)))
        return logits, cache

    def __getattr__(self, k):
        return getattr(self.model, k)
