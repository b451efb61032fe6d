"""The C-ABI library: loads, exports every symbol include/p3v.h declares, and the
ctypes signatures cover exactly that set.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

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
    assert _lib.lib().p3v_version() == 600
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
            decl = re.sub(r"^(const\s+)?(void|uint16_t|uint8_t|uint32_t|int32_t|int64_t|float|int)\s*\*?\s*", "", stmt)
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


def test_library_never_allocates_or_reads_the_environment_per_launch():
    """include/p3v.h: launches never allocate or synchronise (callers pass workspaces, e.g. p3v_gemm_ws_bytes); knobs come
    from ONE table filled once (p3v_runtime.hip).  Checked on the sources and on the library's undefined symbols."""
    import subprocess
    csrc = os.path.join(ROOT, "phi-3-vision-mlx_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            src = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
            for banned in ("hipMalloc", "hipFree", "hipDeviceSynchronize", "hipStreamSynchronize", "hipMemcpy("):
                assert banned not in src, (f, banned)
            if f != "p3v_runtime.hip":
                assert "getenv" not in src, f
    from phi_3_vision_mlx_amd import _lib
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for banned in ("hipMalloc", "hipFree", "hipDeviceSynchronize", "hipStreamSynchronize"):
        assert not re.search(rf"\b{banned}\b", und), banned


def test_tuning_table_and_gemm_workspace_size():
    from phi_3_vision_mlx_amd import _lib
    l = _lib.lib()
    v = ctypes.c_int(0)
    assert l.p3v_get_tuning(b"gemm_splitk_max_m", ctypes.byref(v)) == 0 and v.value == 1024
    assert l.p3v_set_tuning(b"no_such_knob", 1) == -22
    assert l.p3v_set_tuning(b"combine_g", 4) == 0 and l.p3v_get_tuning(b"combine_g", ctypes.byref(v)) == 0 and v.value == 4
    assert l.p3v_set_tuning(b"combine_g", -1) == 0
    # K-slice shapes: S slices of fp32 [M, N or 2N]; none elsewhere.  17 .. 256 rows: the 128 x 64-tile kernel (one workgroup per CU:
    # 48 tiles x 4 slices for down_proj, gate_up's 256 tiles in one pass); up to 1024 rows with < 256 tiles: the 128 x 128 kernel's split
    assert l.p3v_gemm_ws_bytes(128, 3072, 8192, _lib.EPI_RESID_BF16) == 4 * 128 * 3072 * 4
    assert l.p3v_gemm_ws_bytes(128, 8192, 3072, _lib.EPI_SILU_MUL) == 0
    assert l.p3v_gemm_ws_bytes(300, 3072, 8192, _lib.EPI_RESID_BF16) == 4 * 300 * 3072 * 4
    assert l.p3v_gemm_ws_bytes(2531, 3072, 8192, _lib.EPI_RESID_BF16) == 0
    assert l.p3v_gemm_ws_bytes(8, 3072, 8192, _lib.EPI_NONE) == 0
    assert l.p3v_gemm_ws_bytes(128, 3072, 8192, _lib.EPI_BIAS) == 0


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("n_split", list(range(13, 25)))
def test_fused_decode_attention_role_placement_is_a_bijection(n_split, mode):
    """p3v_attention_decode_fused_role (round 6): the 1-D grid of the fused attention + o_proj launch.  For every cache capacity
    the fused form takes (13 .. 24 tiles of 128 keys, 32 heads, o_n = 3072): every (head, split) is played by exactly one
    workgroup, every projection unit 0 .. 383 by exactly one, the 32 merging workgroups are the LAST workgroups of virtual CUs
    224 .. 255 (workgroups L, L + 256, L + 512 share a CU) and nobody on those CUs carries a projection unit; every other CU's
    last workgroup carries one (placement 2: also the second ones; placement 1: those sit on second-to-last workgroups)."""
    from phi_3_vision_mlx_amd import _lib
    l = _lib.lib()
    nh, o_n = 32, 3072
    G = nh * n_split
    out = (ctypes.c_int * 4)()
    roles = []
    for wg in range(G):
        assert l.p3v_attention_decode_fused_role(wg, nh, n_split, o_n, mode, out) == 0
        roles.append(tuple(out))
    assert sorted((r[1], r[0]) for r in roles) == [(h, s) for h in range(nh) for s in range(n_split)]
    units = [u for r in roles for u in r[2:] if u >= 0]
    assert sorted(units) == list(range(o_n // 8))
    n_on = [sum(1 for wg in range(v, G, 256)) for v in range(256)]
    for wg, (split, head, u1, u2) in enumerate(roles):
        v, slot = wg % 256, wg // 256
        last = slot == n_on[v] - 1
        if split == n_split - 1:
            assert v == 224 + head and last and u1 < 0 and u2 < 0
        if v >= 224:
            assert u1 < 0 and u2 < 0
        elif last:
            assert u1 == v
        if mode == 2:
            assert (u1 >= 0) == (last and v < 224) and (u2 < 0 or u1 >= 0)
        else:
            assert u2 < 0 and (u1 < 0 or slot >= n_on[v] - 2)
    assert l.p3v_attention_decode_fused_role(0, 32, 12, o_n, mode, out) == _lib.ERR_UNSUPPORTED      # too few workgroups for 384 units
    assert l.p3v_attention_decode_fused_role(0, 8, 16, o_n, mode, out) == _lib.ERR_UNSUPPORTED       # not the 32-head geometry
    assert l.p3v_attention_decode_fused_role(G, nh, n_split, o_n, mode, out) == -22


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "phi-3-vision-mlx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "phi3v_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.skipif(os.environ.get("P3V_ASAN") != "1", reason="opt-in (P3V_ASAN=1): a ~50 s AddressSanitizer host build of libp3v.so")
def test_asan_host_build_of_the_launchers():
    """`-fsanitize=address` HOST build of the extension; its launchers' argument validation, workspace
    sizing and tuning table are driven without a GPU (tools/asan_host_check.cpp).  profiles/r04_asan_host_check.txt holds a run."""
    import subprocess
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_build.sh")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "all launcher argument paths clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
