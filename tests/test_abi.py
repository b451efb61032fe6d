"""The C-ABI library: loads, exports every symbol include/p3v.h declares, and the
ctypes signatures cover exactly that set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "p3v.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(p3v_[a-z0-9_]+)\s*\(", h)))


def test_header_symbols_are_exported_and_bound():
    from phi_3_vision_mlx_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libp3v.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/p3v.h but not exported by libp3v.so"
    assert sorted(_lib.SIGNATURES) == syms, set(_lib.SIGNATURES) ^ set(syms)
    assert _lib.lib().p3v_version() == 100
    assert _lib.lib().p3v_strerror(-22).decode().startswith("invalid argument")


def test_struct_layouts_match_header_field_order():
    from phi_3_vision_mlx_amd import _lib
    h = open(os.path.join(ROOT, "include", "p3v.h")).read()
    for cname, cls in (("p3v_gemm_args_t", _lib.GemmArgs), ("p3v_gemv_args_t", _lib.GemvArgs),
                       ("p3v_attn_args_t", _lib.AttnArgs), ("p3v_attn_decode_args_t", _lib.AttnDecArgs),
                       ("p3v_attn_decode_q8_args_t", _lib.AttnDecQ8Args), ("p3v_gemv_fp8_args_t", _lib.GemvF8Args)):
        end = h.index("} " + cname)
        body = h[h.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            decl = re.sub(r"^(const\s+)?(void|uint16_t|uint8_t|int32_t|float|int)\s*\*?\s*", "", stmt)
            names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
        assert names == [f for f, _ in cls._fields_], (cname, names)


def test_ops_fail_loudly_without_gpu_or_library(monkeypatch):
    import pytest
    import torch
    from phi_3_vision_mlx_amd import _lib, ops
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ops.rmsnorm(torch.zeros(1, 8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16), 1e-5)
        from phi_3_vision_mlx_amd.model import Phi3VModel
        with pytest.raises(RuntimeError):
            Phi3VModel(None, {})
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libp3v.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "phi-3-vision-mlx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "phi3v_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f
