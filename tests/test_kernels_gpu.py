"""Per-kernel parity: each HIP kernel (called through the C ABI) against the
oracle's restatement of the same reference op on the same seeded inputs.

Tolerances: bf16 outputs are compared with rtol 2^-7 (one bf16 ulp is 2^-8
relative) plus a small atol for cancellation; index outputs are exact."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from phi_3_vision_mlx_amd import ops as o
    o.L.lib()
    return o


@pytest.fixture(scope="module")
def orc():
    import phi3v_oracle
    return phi3v_oracle


def g(shape, seed, std=1.0, dtype=BF16):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=gen) * std).to(dtype)


def close(a, b, rtol=2 ** -7, atol=1e-2):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{bad}/{a.numel()} mismatches; max err {err.max().item():.5f} at |ref| {b.abs().max().item():.3f}"


def test_device_props(ops):
    p = ops.device_props(0)
    assert p["arch"].startswith("gfx950"), p
    assert p["wave_size"] == 64 and p["cu_count"] >= 200


@pytest.mark.parametrize("n,H", [(1, 3072), (7, 192), (300, 3072)])
def test_embed_gather(ops, n, H):
    table = g((1000, H), 1)
    ids = torch.randint(-5, 1000, (n,), dtype=torch.int32)
    out = ops.embed_gather(ids.cuda(), table.cuda())
    ref = table[ids.long().clamp_min(0)]
    assert torch.equal(out.cpu(), ref)                     # pure copy: bit exact


@pytest.mark.parametrize("rows,H", [(1, 3072), (5, 192), (257, 3072)])
def test_rmsnorm(ops, orc, rows, H):
    x, w = g((rows, H), 2, 3.0), g((H,), 3, 0.1) + 1
    out = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5)
    ref = orc.rms_norm(x, w, 1e-5)
    # mx.fast.rms_norm rounds the normalised row to bf16 BEFORE the weight multiply (pinned by tests/golden/ref_model_tiny.npz):
    # the kernel reproduces both roundings -- bit-exact but for the rare entry where the device's rsqrt (1 fp32 ulp off
    # torch's) tips the first rounding, which the second one can stretch to two bf16 ulps
    close(out, ref, rtol=2 ** -6, atol=1e-3)
    assert (out.cpu() == ref).float().mean().item() > 0.999


@pytest.mark.parametrize("rows,H,f32", [(3, 1024, False), (577, 1024, True), (9, 128, False)])
def test_layernorm(ops, orc, rows, H, f32):
    x = g((rows, H), 4, 2.0, F32) + 0.5
    w, b = g((H,), 5, 0.1) + 1, g((H,), 6, 0.1)
    out = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5, out_f32=f32)
    ref = orc.layer_norm(x, w, b, 1e-5)
    if f32:
        close(out, ref, rtol=1e-5, atol=1e-5)
    else:
        close(out, ref.to(BF16), atol=1e-3)


EPI_CASES = [
    ("none", 200, 256, 192), ("none", 128, 128, 64), ("none", 1, 9216, 3072), ("none", 333, 32064, 192),
    ("bias", 577, 3072, 1024), ("qgelu", 130, 4096, 1024), ("gelu", 77, 3072, 4096),
    ("resid_f32", 577, 1024, 4096), ("resid_bf16", 300, 3072, 8192), ("silu", 300, 8192, 3072),
    ("silu", 20, 256, 192), ("f32", 65, 192, 128),
    # large-tile (256x256) path: M >= 1024, N % 256 == 0 (N % 128 for silu)
    ("none", 1100, 768, 192), ("bias", 1300, 1024, 128), ("qgelu", 1024, 512, 256), ("gelu", 1030, 256, 128),
    ("resid_f32", 2000, 1024, 192), ("resid_bf16", 2531, 3072, 256), ("silu", 1500, 640, 192), ("f32", 1025, 256, 64),
    # split-K path (17 <= M <= 1024 with fewer than 256 tiles, N % 128 == 0, K % 512 == 0): fp32 partials per K slice + a
    # reduce / epilogue launch -- short prompts ("resid_bf16", 300, 3072, 8192) above takes it too
    ("none", 17, 9216, 3072), ("silu", 40, 8192, 3072), ("resid_bf16", 129, 3072, 3072), ("none", 700, 3072, 8192),
]


@pytest.mark.parametrize("epi,M,N,K", EPI_CASES)
def test_gemm(ops, epi, M, N, K):
    # (conftest pins P3V_GEMM_BIG_ROWS=512, so every M >= 1024 case runs rows [0,512) on the big-tile kernel and the
    #  rest on the small one: both kernels and the row offsets of A / out / resid are exercised on small shapes)
    a = g((M, K), 10)
    nw = 2 * N if epi == "silu" else N
    w = g((nw, K), 11, 1.0 / math.sqrt(K))
    bias = g((N,), 12, 0.5)
    af, wf = a.float(), w.float()
    acc = af @ wf.t()
    if epi == "none":
        out, ref = ops.gemm(a.cuda(), w.cuda()), acc.to(BF16)
    elif epi == "bias":
        out, ref = ops.gemm(a.cuda(), w.cuda(), ops.EPI_BIAS, bias=bias.cuda()), (acc + bias.float()).to(BF16)
    elif epi == "qgelu":
        v = acc + bias.float()
        out, ref = ops.gemm(a.cuda(), w.cuda(), ops.EPI_BIAS_QGELU, bias=bias.cuda()), (v * torch.sigmoid(1.702 * v)).to(BF16)
    elif epi == "gelu":
        v = acc + bias.float()
        out, ref = ops.gemm(a.cuda(), w.cuda(), ops.EPI_BIAS_GELU, bias=bias.cuda()), torch.nn.functional.gelu(v).to(BF16)
    elif epi == "resid_f32":
        r = g((M, N), 13, 1.0, F32)
        rc = r.cuda()
        out = ops.gemm(a.cuda(), w.cuda(), ops.EPI_BIAS_RESID_F32, bias=bias.cuda(), resid=rc, out=rc)
        ref = r + acc + bias.float()
        close(out, ref, rtol=1e-3, atol=2e-3)
        return
    elif epi == "resid_bf16":
        r = g((M, N), 13)
        rc = r.cuda()
        out = ops.gemm(a.cuda(), w.cuda(), ops.EPI_RESID_BF16, resid=rc, out=rc)
        ref = (r.float() + acc.to(BF16).float()).to(BF16)
    elif epi == "silu":
        gate, up = acc[:, :N].to(BF16), acc[:, N:].to(BF16)
        out, ref = ops.gemm(a.cuda(), w.cuda(), ops.EPI_SILU_MUL), (gate * torch.sigmoid(gate)) * up
    elif epi == "f32":
        out = ops.gemm(a.cuda(), w.cuda(), ops.EPI_F32)
        close(out, acc, rtol=1e-3, atol=2e-3)
        return
    close(out, ref, rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("epi,M,N,K", [("none", 2531, 9216, 3072), ("silu", 2531, 8192, 3072), ("resid_bf16", 2531, 3072, 3072),
                                       ("resid_f32", 9809, 1024, 1024), ("bias", 9809, 3072, 1024), ("none", 4096, 4096, 512)])
def test_gemm_round_packing(ops, epi, M, N, K):
    """The bench request's prefill shapes with the row split p3v_gemm's round-packing model picks (no pin), then all-big
    and all-small tiles: the three must agree bit for bit (same K order, same epilogue arithmetic per element)."""
    a = g((M, K), 30).cuda()
    w = g(((2 * N if epi == "silu" else N), K), 31, 1.0 / math.sqrt(K)).cuda()
    bias = g((N,), 32, 0.5).cuda()
    r = g((M, N), 33, 1.0, F32 if epi == "resid_f32" else BF16).cuda()

    def run():
        if epi == "none": return ops.gemm(a, w)
        if epi == "silu": return ops.gemm(a, w, ops.EPI_SILU_MUL)
        if epi == "bias": return ops.gemm(a, w, ops.EPI_BIAS, bias=bias)
        rc = r.clone()
        if epi == "resid_bf16": return ops.gemm(a, w, ops.EPI_RESID_BF16, resid=rc, out=rc)
        return ops.gemm(a, w, ops.EPI_BIAS_RESID_F32, bias=bias, resid=rc, out=rc)

    pinned = ops.set_tuning("gemm_big_rows", -1)        # -1: the launcher's own cost model
    try:
        picked = run()
        ops.set_tuning("gemm_big_rows", 0)
        small = run()
        ops.set_tuning("gemm_big_rows", 1000000)
        big = run()
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("gemm_big_rows", pinned)
    assert torch.equal(picked, small) and torch.equal(picked, big)
    acc = a[:64].float() @ w.float().t()          # spot check of the first rows against fp32 torch
    if epi == "none": close(picked[:64], acc.to(BF16), rtol=2 ** -6, atol=2e-2)


SKINNY_CASES = [("none", 128, 9216, 3072), ("none", 17, 9216, 3072), ("resid_bf16", 128, 3072, 3072), ("resid_bf16", 100, 3072, 8192),
                ("silu", 128, 8192, 3072), ("silu", 33, 8192, 3072), ("silu", 256, 1024, 3072), ("none", 129, 3072, 3072),
                ("resid_bf16", 250, 192, 768), ("silu", 64, 96, 384), ("none", 200, 64, 64), ("resid_bf16", 40, 3072, 8192), ("none", 64, 9216, 3072)]


@pytest.mark.parametrize("epi,M,N,K", SKINNY_CASES)
def test_gemm_skinny_rows(ops, epi, M, N, K):
    """17 .. 256 rows run on the 128 x 64-tile weight-streaming kernel (p3v_gemm_skinny.hip), in one pass or as K slices + the
    reduction launch: against a float64 product of the same bf16 operands with the reference's roundings (Linear output to bf16,
    then the elementwise ops), and against the 128 x 128-tile path the shape took before (same roundings, another fp32 summation
    order: equal but for last-place flips of the bf16 rounding)."""
    EPI = {"none": ops.EPI_NONE, "resid_bf16": ops.EPI_RESID_BF16, "silu": ops.EPI_SILU_MUL}[epi]
    assert ops.L.lib().p3v_gemm_ws_bytes(M, N, K, EPI) >= 0
    a = g((M, K), 60).cuda()
    w = g(((2 * N if epi == "silu" else N), K), 61, 1.0 / math.sqrt(K)).cuda()
    r = g((M, N), 62).cuda() if epi == "resid_bf16" else None
    out = ops.gemm(a, w, EPI, resid=r)
    old = ops.set_tuning("gemm_no_skinny", 1)
    try:
        other = ops.gemm(a, w, EPI, resid=r)
    finally:
        ops.set_tuning("gemm_no_skinny", old)
    z = (a.double() @ w.double().t())
    if epi == "silu":
        gate, up = z[:, :N].to(BF16).double(), z[:, N:].to(BF16).double()
        ref = (gate * torch.sigmoid(gate)).to(BF16).double() * up
    elif epi == "resid_bf16":
        ref = r.double() + z.to(BF16).double()
    else:
        ref = z
    close(out, ref.to(BF16), rtol=2 ** -6, atol=2e-2)
    close(out, other, rtol=2 ** -6, atol=2e-2)
    assert (out == other).float().mean().item() > 0.99
    if M <= 64:                                                 # <= 64 rows run on 64-row tiles: the same sums as the 128-row tiles, bit for bit
        old = ops.set_tuning("gemm_skinny_tm128", 1)
        try:
            assert torch.equal(ops.gemm(a, w, EPI, resid=r), out)
        finally:
            ops.set_tuning("gemm_skinny_tm128", old)
    for s_pin in (1, 2):                                        # the split pinned: one pass / two slices, same values up to summation order
        if K % (s_pin * 64) == 0:
            old = ops.set_tuning("gemm_skinny_s", s_pin)
            try:
                close(ops.gemm(a, w, EPI, resid=r), out, rtol=2 ** -6, atol=2e-2)
            finally:
                ops.set_tuning("gemm_skinny_s", old)


ROWS_SHAPES = [("none", 9216, 3072), ("resid_bf16", 3072, 3072), ("silu", 8192, 3072), ("resid_bf16", 3072, 8192), ("none", 32064, 3072)]


@pytest.mark.parametrize("M", [9, 15, 16, 17, 24, 32])
@pytest.mark.parametrize("epi,N,K", ROWS_SHAPES)
def test_gemm_rows_9_to_32(ops, epi, N, K, M):
    """9 .. 32 rows on the register-streaming kernel (p3v_gemm_rows.hip, round 6: weights HBM -> registers in full lines, x as resident
    MFMA B fragments, K quarters per wave; qkv / gate_up / lm_head in one pass, o_proj / down as 4 K slices + the reduction launch)
    on the decoder's own shapes: against a float64 product with the reference's roundings, against the 64-row-tile kernel the shape
    took before (same roundings, another summation order), three launches bit-identical, rows beyond M untouched."""
    EPI = {"none": ops.EPI_NONE, "resid_bf16": ops.EPI_RESID_BF16, "silu": ops.EPI_SILU_MUL}[epi]
    assert ops.L.lib().p3v_gemm_rows_slices(M, N, K, EPI) in (1, 4)
    a = g((M, K), 160 + M).cuda()
    w = g(((2 * N if epi == "silu" else N), K), 161, 1.0 / math.sqrt(K)).cuda()
    r = g((M, N), 162).cuda() if epi == "resid_bf16" else None
    buf = torch.full((M + 3, N), 7.0, dtype=BF16, device="cuda")            # three guard rows behind the output
    out = ops.gemm(a, w, EPI, resid=r, out=buf[:M])
    assert bool((buf[M:] == 7.0).all())
    for _ in range(2):
        assert torch.equal(ops.gemm(a, w, EPI, resid=r), out)
    old = ops.set_tuning("gemm_rows", 0)
    try:
        other = ops.gemm(a, w, EPI, resid=r)
    finally:
        ops.set_tuning("gemm_rows", old)
    z = (a.double() @ w.double().t())
    if epi == "silu":
        gate, up = z[:, :N].to(BF16).double(), z[:, N:].to(BF16).double()
        ref = (gate * torch.sigmoid(gate)).to(BF16).double() * up
    elif epi == "resid_bf16":
        ref = r.double() + z.to(BF16).double()
    else:
        ref = z
    close(out, ref.to(BF16), rtol=2 ** -6, atol=2e-2)
    close(out, other, rtol=2 ** -6, atol=2e-2)
    assert (out == other).float().mean().item() > 0.99


def test_gemm_rows_declines_other_shapes(ops):
    """Not this kernel's: 8 rows and fewer (k_gemv_mfma8), more than 32, a K it has no slicing for, a bias epilogue; those keep their
    previous kernels and results."""
    lib = ops.L.lib()
    assert lib.p3v_gemm_rows_slices(8, 9216, 3072, ops.EPI_NONE) == 0 and lib.p3v_gemm_rows_slices(33, 9216, 3072, ops.EPI_NONE) == 0
    assert lib.p3v_gemm_rows_slices(16, 9216, 1024, ops.EPI_NONE) == 0 and lib.p3v_gemm_rows_slices(16, 1024, 3072, ops.EPI_BIAS) == 0
    a, w = g((16, 1024), 170).cuda(), g((512, 1024), 171, 1 / 32).cuda()
    close(ops.gemm(a, w), (a.double() @ w.double().t()).to(BF16), rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(128, 3072, 3072), (100, 3072, 8192), (17, 3072, 3072), (250, 192, 768), (129, 1024, 1024), (16, 3072, 8192), (32, 3072, 3072)])
def test_gemm_resid_norm_is_the_two_calls(ops, M, N, K):
    """p3v_gemm_resid_norm == p3v_gemm(P3V_EPI_RESID_BF16) + p3v_rmsnorm, bit for bit, from the projection's own launches (the K-slice
    GEMM + one reduction that also normalises); shapes the library does not split report "unsupported" and launch nothing."""
    a, w = g((M, K), 70).cuda(), g((N, K), 71, 1.0 / math.sqrt(K)).cuda()
    r, nw = g((M, N), 72).cuda(), (g((N,), 73, 0.1) + 1).cuda()
    two = ops.gemm(a, w, ops.EPI_RESID_BF16, resid=r)
    two_n = ops.rmsnorm(two, nw, 1e-5)
    out, normed = torch.full_like(two, 7.0), torch.full_like(two, 7.0)
    assert ops.gemm_resid_norm(a, w, r, nw, 1e-5, normed, out=out)
    assert torch.equal(out, two) and torch.equal(normed, two_n)
    x = r.clone()                                               # in place on the residual stream, as the decoder layer calls it
    assert ops.gemm_resid_norm(a, w, x, nw, 1e-5, normed, out=x)
    assert torch.equal(x, two) and torch.equal(normed, two_n)
    # not a split shape (too many rows; a wide N that runs in one pass): nothing is launched
    big = g((300, K), 74).cuda()
    sentinel = torch.full((300, N), 7.0, dtype=BF16, device="cuda")
    assert not ops.gemm_resid_norm(big, w, g((300, N), 75).cuda(), nw, 1e-5, sentinel.clone(), out=sentinel)
    torch.cuda.synchronize()
    assert bool((sentinel == 7.0).all())


@pytest.mark.parametrize("epi,M,N,K", [("resid_bf16", 129, 3072, 8192), ("silu", 40, 1024, 3072), ("none", 300, 9216, 3072)])
def test_gemm_splitk_workspace_is_the_callers(ops, epi, M, N, K):
    """p3v_gemm never allocates: the split-K partials live in a workspace the caller sizes with p3v_gemm_ws_bytes.  With the
    workspace the call is capturable in a hipGraph (replay == eager, bit for bit); without one the same shape runs on the
    one-pass kernel and agrees up to fp32 summation order."""
    import ctypes as C
    L = ops.L
    EPI = {"none": ops.EPI_NONE, "resid_bf16": ops.EPI_RESID_BF16, "silu": ops.EPI_SILU_MUL}[epi]
    a = g((M, K), 50).cuda()
    w = g(((2 * N if epi == "silu" else N), K), 51, 1.0 / math.sqrt(K)).cuda()
    r = g((M, N), 52).cuda() if epi == "resid_bf16" else None
    need = L.lib().p3v_gemm_ws_bytes(M, N, K, EPI)
    assert need > 0
    eager = ops.gemm(a, w, EPI, resid=r)

    def raw(ws, out):
        args = L.GemmArgs(a.data_ptr(), w.data_ptr(), out.data_ptr(), 0, 0 if r is None else r.data_ptr(), 0, M, N, K, K, K, N, EPI, 0,
                          0 if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel())
        L.check(L.lib().p3v_gemm(C.byref(args), torch.cuda.current_stream().cuda_stream), "gemm")
        return out

    no_ws = raw(None, torch.empty_like(eager))
    short = raw(torch.empty(need - 16, dtype=torch.uint8, device="cuda"), torch.empty_like(eager))     # too small: one-pass kernel
    torch.cuda.synchronize()
    assert torch.equal(no_ws, short)
    close(no_ws, eager, rtol=2 ** -6, atol=2e-2)
    ref = a.float() @ w.float().t()
    if epi == "none": close(eager, ref.to(BF16), rtol=2 ** -6, atol=2e-2)

    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = torch.zeros_like(eager)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = ops.Graph()
        graph.begin()
        raw(ws, out)
        graph.end()
        graph.launch()
    side.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("M,N,K,epi", [(2531, 9216, 3072, "none"), (2531, 3072, 8192, "resid"), (2531, 8192, 3072, "silu"), (9809, 4096, 1024, "none"),
                                       (300, 3072, 3072, "resid"), (1, 9216, 3072, "gemv"), (8, 3072, 8192, "gemv")])
def test_matrix_kernels_are_deterministic(ops, M, N, K, epi):
    """Ten launches of the prompt-sized GEMMs (both tile sizes, round packing, split-K) and of the skinny GEMVs on the same inputs
    give the same bits.  (The attention kernels have their own, test_attention_prefill_kernels_are_deterministic: that is where
    a launch-to-launch difference was once found.)"""
    x = g((M, K), 70).cuda()
    w = (g(((2 * N if epi == "silu" else N), K), 71) * 0.05).cuda()
    res = g((M, N), 72).cuda()
    first = None
    for rep in range(10):
        if epi == "gemv":
            out = ops.gemv(x, w, ops.EPI_NONE)
        elif epi == "resid":
            out = ops.gemm(x, w, ops.EPI_RESID_BF16, resid=res)
        elif epi == "silu":
            out = ops.gemm(x, w, ops.EPI_SILU_MUL)
        else:
            out = ops.gemm(x, w, ops.EPI_NONE)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            assert torch.isfinite(first.float()).all()
        else:
            assert torch.equal(out.view(torch.int16), first.view(torch.int16)), f"launch {rep} differs"


def test_gemm_asymmetric_identity(ops):
    """A = I against an asymmetric W: catches a transposed C write (guide rule 16)."""
    K = 128
    a = torch.eye(K, dtype=BF16)
    w = (torch.arange(192 * K, dtype=F32).reshape(192, K) % 251 / 16).to(BF16)
    out = ops.gemm(a.cuda(), w.cuda())
    assert torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize("M,N,K,epi", [(1, 9216, 3072, "none"), (1, 3072, 8192, "resid"), (1, 8192, 3072, "silu"),
                                       (3, 576, 192, "none"), (8, 3072, 8192, "resid"), (5, 256, 192, "silu"),
                                       (1, 32064, 3072, "norm"), (8, 9216, 3072, "norm"), (2, 101, 192, "none"),
                                       (16, 3072, 3072, "resid"), (16, 8192, 3072, "silu"), (12, 9216, 3072, "norm"),
                                       (2, 3072, 8192, "resid"), (9, 32064, 3072, "norm"), (16, 1000, 1024, "none"),
                                       # 5 <= M <= 8: activations staged in LDS (k_gemv_mfma8), 4 waves (K <= 3072) / 8 waves
                                       (8, 8192, 3072, "silu"), (6, 3072, 3072, "resid"), (7, 32064, 3072, "norm"),
                                       (5, 1000, 1024, "none"), (8, 3072, 4096, "none"), (5, 8192, 2048, "silu"),
                                       (6, 100, 1024, "silu"), (7, 37, 2048, "none"), (8, 24, 3072, "resid")])   # ragged N
def test_gemv(ops, orc, M, N, K, epi):
    x = g((M, K), 20)
    nw = 2 * N if epi == "silu" else N
    w = g((nw, K), 21, 1.0 / math.sqrt(K))
    acc = x.float() @ w.float().t()
    if epi == "none":
        out, ref = ops.gemv(x.cuda(), w.cuda()), acc.to(BF16)
    elif epi == "resid":
        r = g((M, N), 22)
        rc = r.cuda()
        out = ops.gemv(x.cuda(), w.cuda(), ops.EPI_RESID_BF16, resid=rc, out=rc)
        ref = (r.float() + acc.to(BF16).float()).to(BF16)
    elif epi == "silu":
        gate, up = acc[:, :N].to(BF16), acc[:, N:].to(BF16)
        out, ref = ops.gemv(x.cuda(), w.cuda(), ops.EPI_SILU_MUL), (gate * torch.sigmoid(gate)) * up
    elif epi == "norm":
        nw_ = g((K,), 23, 0.1) + 1
        out = ops.gemv(x.cuda(), w.cuda(), norm_w=nw_.cuda(), norm_eps=1e-5)
        ref = (orc.rms_norm(x, nw_, 1e-5).float() @ w.float().t()).to(BF16)
    close(out, ref, rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("M,N,K,epi", [(1, 9216, 3072, "norm"), (1, 3072, 8192, "resid"), (1, 8192, 3072, "silu"),
                                       (8, 3072, 3072, "resid"), (16, 8192, 3072, "silu"), (5, 9216, 3072, "norm"),
                                       (2, 3072, 8192, "none"), (1, 32064, 3072, "norm"),
                                       # 5 <= M <= 8: k_gemv_mfma8_f8 (whole-line fragment loads, x staged in LDS)
                                       (8, 8192, 3072, "silu"), (7, 3072, 8192, "resid"), (6, 32064, 3072, "norm"), (8, 1000, 3072, "none")])
def test_gemv_fp8(ops, orc, M, N, K, epi):
    """fp8 weight-only projection == the bf16 math on the DEQUANTISED weights (the only new error is weight rounding,
    which the reference comparison below does not see: both sides use fp8*scale)."""
    x = g((M, K), 24)
    rows = 2 * N if epi == "silu" else N
    w = g((rows, K), 25, 1.0 / math.sqrt(K))
    w8, sc = ops.quantize_fp8_rows(w.cuda())
    wd = ops.dequant_fp8(w8, sc).cpu()
    assert torch.equal(wd.float(), (w8.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]).to(BF16).float())
    assert (wd.float() - w.float()).abs().max() <= w.float().abs().max() * 2 ** -3       # e4m3: 3 mantissa bits
    wdq = w8.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]                   # exact fp32 dequant
    kw = {}
    xin = x
    if epi == "norm":
        nw_ = g((K,), 26, 0.1) + 1
        kw = dict(norm_w=nw_.cuda(), norm_eps=1e-5)
        xin = orc.rms_norm(x, nw_, 1e-5)
    acc = xin.float() @ wdq.t()
    if epi in ("none", "norm"):
        out, ref = ops.gemv_fp8(x.cuda(), w8, sc, **kw), acc.to(BF16)
    elif epi == "resid":
        r = g((M, N), 27)
        rc = r.cuda()
        out = ops.gemv_fp8(x.cuda(), w8, sc, ops.EPI_RESID_BF16, resid=rc, out=rc)
        ref = (r.float() + acc.to(BF16).float()).to(BF16)
    else:
        gate, up = acc[:, :N].to(BF16), acc[:, N:].to(BF16)
        out, ref = ops.gemv_fp8(x.cuda(), w8, sc, ops.EPI_SILU_MUL), (gate * torch.sigmoid(gate)) * up
    close(out, ref, rtol=2 ** -6, atol=3e-2)


@pytest.mark.parametrize("L,T,past", [(5, 40, 7), (70, 128, 8), (100, 192, 3), (64, 64, 0)])
def test_rope_table_and_append(ops, orc, L, T, past):
    from phi_3_vision_mlx_amd.config import make_config, rope_scaling_factor
    cfg = make_config()
    B, nh, nkv, hd = 2, 4, 4, 96
    pids = torch.stack([torch.arange(T), torch.cat([torch.ones(3, dtype=torch.long), torch.arange(T - 3)])])
    cos_ref, sin_ref = orc.su_rope_tables(cfg, T, pids[:, :12])
    inv = 1.0 / (torch.tensor(cfg.rope_scaling["short_factor"], dtype=F32) * (10000.0 ** (torch.arange(0, hd, 2, dtype=F32) / hd)))
    cos, sin = ops.rope_table(pids.float().reshape(-1).cuda(), inv.cuda(), rope_scaling_factor(cfg))
    close(cos.view(B, T, 48), cos_ref[:, 0, :, :48], rtol=1e-5, atol=2e-5)
    close(sin.view(B, T, 48), sin_ref[:, 0, :, :48], rtol=1e-5, atol=2e-5)
    qkv = g((B * L, (nh + 2 * nkv) * hd), 30)
    q = torch.zeros((B, nh, L, hd), dtype=BF16).cuda()
    kc = torch.zeros((B, nkv, T, hd), dtype=BF16).cuda()
    vc = torch.zeros((B, nkv, hd, T), dtype=BF16).cuda()                     # V^T layout
    ops.rope_kv_append(qkv.cuda(), cos, sin, q, kc, vc, B, L, nh, nkv, hd, past, T, True, T, 1)
    x = qkv.view(B, L, nh + 2 * nkv, hd).transpose(1, 2)                     # [B, heads, L, hd]
    cs, sn = cos_ref[:, :, past:past + L], sin_ref[:, :, past:past + L]
    close(q, orc.rotate_half(x[:, :nh], cs, sn).to(BF16), atol=1e-2)
    close(kc[:, :, past:past + L], orc.rotate_half(x[:, nh:nh + nkv], cs, sn).to(BF16), atol=1e-2)
    assert torch.equal(vc[:, :, :, past:past + L].cpu().transpose(2, 3), x[:, nh + nkv:])
    assert vc[:, :, :, :past].abs().sum().item() == 0
    assert kc[:, :, :past].abs().sum().item() == 0 and kc[:, :, past + L:].abs().sum().item() == 0
    # q_scale: queries (only) are multiplied before their ONE rounding to bf16; keys and values are untouched
    qs = (hd ** -0.5) * ops.Q_PRESCALE
    q2, kc2, vc2 = torch.zeros_like(q), torch.zeros_like(kc), torch.zeros_like(vc)
    ops.rope_kv_append(qkv.cuda(), cos, sin, q2, kc2, vc2, B, L, nh, nkv, hd, past, T, True, T, 1, q_scale=qs)
    close(q2, (orc.rotate_half(x[:, :nh], cs, sn) * qs).to(BF16), atol=2e-3)
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc)
    # the plain head split (null tables: the ViT's q / k / v, phi.py:147) with and without q_scale: q = bf16(q * q_scale), bit for bit
    q3, kc3, vc3 = torch.zeros_like(q), torch.zeros_like(kc), torch.zeros_like(vc)
    ops.rope_kv_append(qkv.cuda(), None, None, q3, kc3, vc3, B, L, nh, nkv, hd, past, T, True)
    assert torch.equal(q3.cpu(), x[:, :nh]) and torch.equal(kc3[:, :, past:past + L].cpu(), x[:, nh:nh + nkv])
    q4 = torch.zeros_like(q)
    ops.rope_kv_append(qkv.cuda(), None, None, q4, kc3, vc3, B, L, nh, nkv, hd, past, T, True, q_scale=qs)
    assert torch.equal(q4.cpu(), (x[:, :nh].float() * np.float32(qs)).to(BF16))
    assert torch.equal(kc3[:, :, past:past + L].cpu(), x[:, nh:nh + nkv]) and torch.equal(vc3, vc)


def _attn_ref(orc, q, k, v, scale, allowed):
    w = (q.float() * scale) @ k.float().transpose(-1, -2)
    return (orc.masked_softmax(w, allowed) @ v.float())


@pytest.mark.parametrize("B,L,past,hd,nh,causal,pads", [
    (1, 130, 0, 96, 2, True, None), (2, 70, 0, 96, 3, True, [0, 9]), (2, 577, 0, 64, 2, False, None),
    (1, 1, 300, 96, 4, True, None), (3, 1, 77, 96, 2, True, [0, 5, 70]), (2, 6, 130, 96, 2, True, [3, 0]),
    (1, 16, 64, 96, 2, True, None), (1, 33, 100, 96, 1, True, None)])
def test_attention(ops, orc, B, L, past, hd, nh, causal, pads):
    T = past + L
    Tp = (T + 63) // 64 * 64
    q, k, v = g((B, nh, L, hd), 40), g((B, nh, T, hd), 41), g((B, nh, T, hd), 42)
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
    kc = torch.zeros((B, nh, Tp, hd), dtype=BF16)
    vc = torch.zeros((B, nh, hd, Tp), dtype=BF16)                            # V^T cache layout
    kc[:, :, :T], vc[:, :, :, :T] = k, v.transpose(2, 3)
    kc, vc = kc.cuda(), vc.cuda()
    n_split, ws = 0, None
    if L <= 16:
        n_split = 4
        ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    ops.attention(q.cuda(), out, B, L, nh, nh, hd, hd ** -0.5, causal, past=past, k_past=kc, v_past=vc, past_t=Tp,
                  pad_len=pad.cuda() if pads else None, ws=ws, n_split=n_split, new_is_cache=True)
    t = torch.arange(T)[None, None, None, :]
    qpos = (past + torch.arange(L))[None, None, :, None]
    allowed = (t >= pad[:, None, None, None]) & (qpos >= pad[:, None, None, None])
    if causal:
        allowed = allowed & (t <= qpos)
    allowed = allowed.expand(B, 1, L, T)
    ref = _attn_ref(orc, q, k, v, hd ** -0.5, allowed).transpose(1, 2).reshape(B, L, nh * hd)
    close(out, ref, rtol=2 ** -6, atol=2e-2)


def _prefill_case(ops, orc, B, L, past, hd, nh, causal, pads, prescaled, pp, spike=None, seed=40, q_std=1.0, il_waves=None):
    """Prompt-sized attention through p3v_attention with the kernel pinned (pp = 2: k_attn_prefill_il [pre-scaled q only],
    1: k_attn_prefill_pp, 0: k_attn_prefill_dma) vs the oracle's fp32 attention.  prescaled: q goes in multiplied by scale * log2(e) and rounded once (what
    p3v_rope_kv_append's q_scale produces); the reference then uses exactly those bf16 values divided back in fp32."""
    T = past + L
    Tp = (T + 63) // 64 * 64
    scale = hd ** -0.5
    q, k, v = g((B, nh, L, hd), seed, q_std), g((B, nh, T, hd), seed + 1), g((B, nh, T, hd), seed + 2)
    if spike is not None:                      # one key that beats every other score of ONE query by far: the reference of
        qb, qr, kt = spike                     # that query's exponent jumps by much more than 2^8 in the middle of the walk
        k[qb, :, kt] = q[qb, :, qr] * 6.0
    q_in = (q.float() * (scale * ops.Q_PRESCALE)).to(BF16) if prescaled else q
    q_ref = q_in.float() / (scale * ops.Q_PRESCALE) if prescaled else q.float()
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16).cuda()
    kc = torch.zeros((B, nh, Tp, hd), dtype=BF16)
    vc = torch.zeros((B, nh, hd, Tp), dtype=BF16)
    kc[:, :, :T], vc[:, :, :, :T] = k, v.transpose(2, 3)
    if pp == 2 and not prescaled:
        pytest.skip("the interleaved kernel takes pre-scaled q only (the launcher never picks it otherwise)")
    old, old_il = ops.set_tuning("attn_pp", min(pp, 1)), ops.set_tuning("attn_il", int(pp == 2))
    old_w = ops.set_tuning("attn_il_waves", (8, 4)[seed & 1] if il_waves is None else il_waves)    # both workgroup sizes of the il kernel get exercised
    try:
        ops.attention(q_in.cuda(), out, B, L, nh, nh, hd, scale, causal, past=past, k_past=kc.cuda(), v_past=vc.cuda(), past_t=Tp,
                      pad_len=pad.cuda() if pads else None, new_is_cache=True, q_prescaled=prescaled)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("attn_pp", old), ops.set_tuning("attn_il", old_il), ops.set_tuning("attn_il_waves", old_w)
    t = torch.arange(T)[None, None, None, :]
    qpos = (past + torch.arange(L))[None, None, :, None]
    allowed = (t >= pad[:, None, None, None]) & (qpos >= pad[:, None, None, None])
    if causal:
        allowed = allowed & (t <= qpos)
    allowed = allowed.expand(B, 1, L, T)
    ref = _attn_ref(orc, q_ref, k, v, scale, allowed).transpose(1, 2).reshape(B, L, nh * hd)
    valid = (qpos >= pad[:, None, None, None]).expand(B, 1, L, 1).reshape(B, L, 1).expand(B, L, nh * hd)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert (got[~valid] == 0).all()                                     # a query that is itself padding outputs 0 (Q7)
    close(got[valid], ref[valid], rtol=2 ** -6, atol=2e-2)
    return got


@pytest.mark.parametrize("pp", [2, 1, 0], ids=["interleaved", "pingpong", "dma"])
@pytest.mark.parametrize("prescaled", [True, False], ids=["prescaled", "plain"])
@pytest.mark.parametrize("B,L,past,hd,nh,causal,pads", [
    (1, 700, 0, 96, 2, True, None),            # 3 query blocks of 256, the last one ragged (188 rows: two idle waves)
    (2, 300, 0, 96, 2, True, [0, 77]),         # left padding inside the first tile; pad query rows
    (1, 257, 0, 96, 1, True, None),            # second block holds ONE query
    (1, 40, 200, 96, 2, True, None),           # cached call: 40 new queries over 200 past keys (constrain's long steps)
    (2, 33, 100, 96, 1, True, [3, 100]),       # row 1: every past key is padding
    (2, 577, 0, 64, 2, False, None),           # CLIP: no mask, head_dim 64, 577 = 2 x 256 + 65
    (1, 1100, 0, 96, 1, True, None),           # 18 tiles: ring wrap-around many times
    (1, 64, 0, 96, 1, True, None), (1, 17, 0, 96, 2, True, None),      # one tile; fewer queries than one wave
])
def test_attention_prefill_kernels(ops, orc, B, L, past, hd, nh, causal, pads, prescaled, pp):
    _prefill_case(ops, orc, B, L, past, hd, nh, causal, pads, prescaled, pp, il_waves=8)
    if pp == 2:
        _prefill_case(ops, orc, B, L, past, hd, nh, causal, pads, prescaled, pp, il_waves=4)      # 128-query workgroups


@pytest.mark.parametrize("pp", [2, 1, 0], ids=["interleaved", "pingpong", "dma"])
def test_attention_prefill_long_context_spot_rows(ops, orc, pp):
    """12288 queries x 12288 keys x 8 heads (K / V^T far larger than the L2: the LDS-DMA tiles arrive LATE, which is what
    exposes a missing wait or a ring slot reused too early -- the first version of the ping-pong kernel passed every
    small-shape test and failed here).  64 query rows spread over the length, all heads, against the fp32 oracle formula."""
    B, L, nh, hd = 1, 12288, 8, 96
    scale = hd ** -0.5
    gen = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn((B, nh, L, hd), device="cuda", generator=gen).to(BF16)
    k = torch.randn((B, nh, L, hd), device="cuda", generator=gen).to(BF16)
    v = torch.randn((B, nh, L, hd), device="cuda", generator=gen).to(BF16)
    q_in = (q.float() * (scale * ops.Q_PRESCALE)).to(BF16)
    vt = v.transpose(2, 3).contiguous()
    out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16, device="cuda")
    old, old_il = ops.set_tuning("attn_pp", min(pp, 1)), ops.set_tuning("attn_il", int(pp == 2))
    try:
        for _ in range(3):                                              # races are timing-dependent: a few launches
            ops.attention(q_in, out, B, L, nh, nh, hd, scale, True, k_past=k, v_past=vt, past_t=L, new_is_cache=True, q_prescaled=True)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("attn_pp", old), ops.set_tuning("attn_il", old_il)
    assert torch.isfinite(out.float()).all()
    rows = torch.cat([torch.arange(0, L, 197), torch.tensor([255, 256, 4095, 4096, L - 257, L - 1])]).unique()
    qr = (q_in[0, :, rows].float().cpu() / (scale * ops.Q_PRESCALE))   # [nh, n, hd]
    kf, vf = k[0].float().cpu(), v[0].float().cpu()
    w = (qr * scale) @ kf.transpose(1, 2)                               # [nh, n, L]
    w = w.masked_fill(torch.arange(L)[None, None, :] > rows[None, :, None], float("-inf"))
    ref = (torch.softmax(w, dim=-1) @ vf).transpose(0, 1).reshape(len(rows), nh * hd)
    close(out[0, rows].float().cpu(), ref, rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("B,L,nh,hd,causal", [(17, 577, 16, 64, False), (1, 2531, 32, 96, True), (2, 900, 8, 96, True), (1, 6200, 8, 96, True)])
def test_attention_prefill_kernels_are_deterministic(ops, B, L, nh, hd, causal):
    """Every prompt-sized kernel, plain and pre-scaled Q: ten launches on the same inputs give the same BITS, and they are right.
    (Found the hard way: a softmax whose per-score multiply-add the compiler put into fresh registers -- v_pk_fma_f32 into the
    registers the last S^T MFMA had read -- produced whole 16-query halves wrong by up to 0.8 in a few launches out of ten, at
    the ViT's shape only; one launch per shape, as the parity tests do, passed more often than not.)"""
    gen = torch.Generator(device="cuda").manual_seed(11)
    scale = hd ** -0.5
    q = torch.randn((B, nh, L, hd), device="cuda", generator=gen).to(BF16)
    Tp = (L + 63) // 64 * 64
    k = torch.randn((B, nh, Tp, hd), device="cuda", generator=gen).to(BF16)
    v = torch.randn((B, nh, hd, Tp), device="cuda", generator=gen).to(BF16)
    k[:, :, L:], v[:, :, :, L:] = 0, 0
    rows = torch.arange(0, L, max(1, L // 40), device="cuda")
    w = (q[:, :, rows].float() * scale) @ k[:, :, :L].float().transpose(-1, -2)
    if causal:
        w = w.masked_fill(torch.arange(L, device="cuda")[None, None, None, :] > rows[None, None, :, None], float("-inf"))
    ref = (torch.softmax(w, -1) @ v[:, :, :, :L].float().transpose(-1, -2)).transpose(1, 2).reshape(B, len(rows), nh * hd)
    q_pre = (q.float() * (scale * ops.Q_PRESCALE)).to(BF16)
    kinds = [("dma", 0, 0, 0, False), ("dma", 0, 0, 0, True), ("pingpong", 1, 0, 0, False), ("pingpong", 1, 0, 0, True),
             ("interleaved 8", 1, 1, 8, True), ("interleaved 4", 1, 1, 4, True)]
    saved = [ops.set_tuning(n, 0) for n in ("attn_pp", "attn_il", "attn_il_waves")]
    try:
        for name, pp, il, nw, pre in kinds:
            ops.set_tuning("attn_pp", pp), ops.set_tuning("attn_il", il), ops.set_tuning("attn_il_waves", nw)
            first = None
            for rep in range(10):
                out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16, device="cuda")
                ops.attention(q_pre if pre else q, out, B, L, nh, nh, hd, scale, causal, k_past=k, v_past=v, past_t=Tp, new_is_cache=True,
                              q_prescaled=pre)
                torch.cuda.synchronize()
                if first is None:
                    first = out
                    ref_q = ref if not pre else None
                    if pre:                                    # the reference of the ROUNDED pre-scaled queries
                        wq = (q_pre[:, :, rows].float() / ops.Q_PRESCALE) @ k[:, :, :L].float().transpose(-1, -2)
                        if causal:
                            wq = wq.masked_fill(torch.arange(L, device="cuda")[None, None, None, :] > rows[None, None, :, None], float("-inf"))
                        ref_q = (torch.softmax(wq, -1) @ v[:, :, :, :L].float().transpose(-1, -2)).transpose(1, 2).reshape(B, len(rows), nh * hd)
                    close(out[:, rows].float().cpu(), ref_q.cpu(), rtol=2 ** -6, atol=2e-2)
                else:
                    n_diff = int((out.view(torch.int16) != first.view(torch.int16)).sum())
                    assert n_diff == 0, f"{name} (pre-scaled {pre}): launch {rep} differs from launch 0 in {n_diff} output words"
    finally:
        for n, val in zip(("attn_pp", "attn_il", "attn_il_waves"), saved):
            ops.set_tuning(n, val)


@pytest.mark.parametrize("seed", range(10))
def test_attention_prefill_interleaved_random_shapes(ops, orc, seed):
    """k_attn_prefill_il over seeded random shapes (the launcher only takes it for long single prompts; pinned here it must hold for
    any length): 1..700 new queries over 0..400 cached keys, 1-3 rows with random left padding (sometimes a whole row's past),
    causal or not, head dim 96 / 64, 1-3 heads, modest or large score magnitudes -- first / last tile on neutral operands, idle
    waves, ragged last tile, ring wrap-around and the reference path all get hit in some combination."""
    rng = np.random.default_rng(1000 + seed)
    B, nh, hd = int(rng.integers(1, 4)), int(rng.integers(1, 4)), (96, 64)[int(rng.integers(0, 2))]
    L, past = int(rng.integers(1, 701)), int(rng.integers(0, 2)) * int(rng.integers(1, 401))
    causal = bool(rng.integers(0, 4)) or past > 0
    pads = [int(rng.integers(0, past + L // 2 + 1)) * int(rng.integers(0, 2)) for _ in range(B)] if rng.integers(0, 2) else None
    _prefill_case(ops, orc, B, L, past, hd, nh, causal, pads, True, 2, seed=200 + seed, q_std=(1.0, 4.0)[int(rng.integers(0, 2))])


@pytest.mark.parametrize("pp", [2, 1, 0], ids=["interleaved", "pingpong", "dma"])
def test_attention_prefill_reference_jump_and_large_scores(ops, orc, pp):
    """The ping-pong kernel keeps a per-query reference for the exponent and moves it only when a tile's maximum exceeds it
    by more than 2^8 (guide T13 hazard: O, l and the pending P must be rescaled exactly once).  (a) a spiked key in the
    middle of the walk forces that slow path for ONE query while its neighbours stay on the fast path; (b) scores of
    magnitude ~100 (log2 units) with the FIRST tile far below the later ones; (c) the first visible tile sets the reference
    even when its scores are hugely negative (no underflow of the whole row)."""
    _prefill_case(ops, orc, 1, 600, 0, 96, 2, True, None, True, pp, spike=(0, 450, 300))
    _prefill_case(ops, orc, 1, 600, 0, 96, 2, True, None, pp == 2, pp, spike=(0, 599, 64))
    _prefill_case(ops, orc, 1, 400, 0, 96, 1, True, None, True, pp, q_std=6.0, seed=90)
    _prefill_case(ops, orc, 2, 577, 0, 64, 1, False, None, pp == 2, pp, q_std=8.0, seed=91)


@pytest.mark.parametrize("B,L,past,nh,n_split,pads", [(1, 1, 300, 4, 5, None), (2, 1, 63, 2, 1, [0, 7]), (1, 6, 130, 2, 3, None),
                                                      (3, 1, 2000, 2, 32, [0, 100, 1999]), (1, 16, 0, 2, 2, None),
                                                      (2, 4, 61, 2, 2, [0, 7]), (2, 5, 20, 2, 1, [3, 0]), (1, 3, 700, 2, 3, None),
                                                      (2, 4, 126, 2, 3, [0, 9]), (1, 2, 31, 2, 1, None)])
@pytest.mark.parametrize("dev_past,fused_merge", [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize("tile", [64, 128])
def test_attention_decode_fused(ops, orc, B, L, past, nh, n_split, pads, dev_past, fused_merge, tile):
    """Fused split + RoPE + KV append + split-KV attention vs the oracle's rotate_half / cache / softmax.
    tile = 128: the cache capacity is a multiple of 128 and there is one split per 128 keys -> k_attn_decode128
    ((2, 4, 61) and (1, 6, 130) put new rows on both sides of a tile / wave-slice boundary)."""
    from phi_3_vision_mlx_amd.config import make_config, rope_scaling_factor
    cfg = make_config()
    hd, T = 96, (past + L + 5 + 63) // 64 * 64
    if tile == 128:
        T = (past + L + 5 + 127) // 128 * 128
        n_split = T // 128
    qkv = g((B * L, 3 * nh * hd), 45)
    kc, vc = g((B, nh, T, hd), 46), g((B, nh, T, hd), 47)
    cos_ref, sin_ref = orc.su_rope_tables(cfg, T, None)
    inv = 1.0 / (torch.tensor(cfg.rope_scaling["short_factor"], dtype=F32) * (10000.0 ** (torch.arange(0, hd, 2, dtype=F32) / hd)))
    pos = torch.arange(T, dtype=F32).repeat(B)
    cos, sin = ops.rope_table(pos.cuda(), inv.cuda(), rope_scaling_factor(cfg))
    pad = torch.tensor(pads if pads else [0] * B, dtype=torch.int32)
    kcc, vcc = kc.cuda(), vc.transpose(2, 3).contiguous().cuda()         # V^T cache layout
    out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    cos_s = torch.empty((B, L, hd // 2), dtype=F32).cuda()
    sin_s = torch.empty_like(cos_s)
    ops.stage_rope(cos.view(B, T, -1), sin.view(B, T, -1), cos_s, sin_s, B, L, T, d_past=d_past)
    assert torch.equal(cos_s.cpu(), cos.view(B, T, -1)[:, past:past + L].cpu())
    ops.attention_decode(qkv.cuda(), cos_s, sin_s, L, kcc, vcc, out, B, L, nh, nh, hd, hd ** -0.5, 0 if dev_past else past, T, ws,
                         n_split, pad_len=pad.cuda() if pads else None, d_past=d_past if dev_past else None, merge_in_launch=fused_merge)
    # the in-launch merge validates partials against a sentinel: every launch (merged in the launch or by the combine
    # kernel) must leave the workspace all-ones, whatever the shape that used it
    assert (ws.view(torch.int32) == -1).all()
    x = qkv.view(B, L, 3 * nh, hd).transpose(1, 2)
    cs, sn = cos_ref[:, :, past:past + L], sin_ref[:, :, past:past + L]
    q = orc.rotate_half(x[:, :nh], cs, sn).to(BF16)
    k_new = orc.rotate_half(x[:, nh:2 * nh], cs, sn).to(BF16)
    v_new = x[:, 2 * nh:]
    kf = torch.cat([kc[:, :, :past], k_new], dim=2)
    vf = torch.cat([vc[:, :, :past], v_new], dim=2)
    t = torch.arange(past + L)[None, None, None, :]
    qpos = (past + torch.arange(L))[None, None, :, None]
    allowed = ((t >= pad[:, None, None, None]) & (qpos >= pad[:, None, None, None]) & (t <= qpos)).expand(B, 1, L, past + L)
    ref = _attn_ref(orc, q, kf, vf, hd ** -0.5, allowed).transpose(1, 2).reshape(B, L, nh * hd)
    close(out, ref, rtol=2 ** -6, atol=2e-2)
    close(kcc[:, :, past:past + L], k_new, atol=1e-2)                       # appended rows
    vback = vcc.cpu().transpose(2, 3)
    assert torch.equal(vback[:, :, past:past + L], v_new)
    assert torch.equal(vback[:, :, :past], vc[:, :, :past]) and torch.equal(vback[:, :, past + L:], vc[:, :, past + L:])
    assert torch.equal(kcc[:, :, :past].cpu(), kc[:, :, :past]) and torch.equal(kcc[:, :, past + L:].cpu(), kc[:, :, past + L:])


@pytest.mark.parametrize("past,cap,dev_past", [(2531, 2688, True), (300, 1664, False), (2559, 2688, True), (1000, 1792, True)])
def test_attention_decode_with_fused_oproj_is_bit_identical_to_two_launches(ops, past, cap, dev_past):
    """k_attn_decode128_o: attention + o_proj + residual in ONE launch (B = L = 1, 32 heads x 96) against the two launches it
    replaces -- p3v_attention_decode then p3v_gemv(P3V_EPI_RESID_BF16): the residual row, the attention output and the appended
    K / V must be BIT-IDENTICAL (the projecting waves repeat k_gemv3's arithmetic exactly), ten launches in a row give the same
    bits, the other output buffer is re-armed (all 0xFFFF) and the split workspace is left all-ones."""
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T, n_split = cap, cap // 128
    assert ops.attention_decode_can_fuse_oproj(B, L, nh, hd, n_split, T, H, True)
    assert not ops.attention_decode_can_fuse_oproj(2, L, nh, hd, n_split, T, H, True)        # B = 1 only
    assert ops.attention_decode_can_fuse_oproj(B, L, nh, hd, n_split, T, H, False)           # (round 6: then the merge launch carries the o_proj)
    qkv = g((1, 3 * nh * hd), 145).cuda()
    kc0, vc0 = g((B, nh, T, hd), 146).cuda(), g((B, nh, hd, T), 147).cuda()
    wo = (g((H, nh * hd), 148) * 0.05).cuda()
    x0 = g((1, H), 149).cuda()
    gen = torch.Generator(device="cuda").manual_seed(150)       # (unseeded tables made this test input-dependent: tools/stress_fused_oproj.py)
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda", generator=gen), torch.rand((B, 1, hd // 2), device="cuda", generator=gen)
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=True)
    hp = past - 40 if dev_past else past                         # (with d_past: a lower bound of the length, not the length)
    # two launches
    k1, v1, o1, x1 = kc0.clone(), vc0.clone(), torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode(qkv, cos, sin, 1, k1, v1, o1, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw)
    ops.gemv(o1.view(1, H), wo, ops.EPI_RESID_BF16, resid=x1, out=x1)
    # one launch, ten times
    for rep in range(10):
        k2, v2, x2 = kc0.clone(), vc0.clone(), x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        ops.attention_decode(qkv, cos, sin, 1, k2, v2, o2, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw,
                             o_proj_w=wo, o_proj_x=x2, o_rearm=other)
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + gemv"
        assert torch.equal(k2, k1) and torch.equal(v2, v1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()
    assert not torch.isnan(x1.float()).any() and (x1.float() - x0.float()).abs().max().item() > 0.01


@pytest.mark.parametrize("q4", [False, True], ids=["bf16", "q4"])
@pytest.mark.parametrize("past,cap,n_split,dev_past", [(7000, 7168, 112, False), (9000, 9216, 24, True), (300, 384, 3, False), (33000, 33280, 24, True)])
def test_merge_launch_with_fused_oproj_is_bit_identical_to_three_launches(ops, past, cap, n_split, dev_past, q4, nkv=32):
    """k_attn_combine_o (round 6): plans whose split-KV partials are merged by a launch of their own (long contexts; 64-key tiles or
    the multi-tile streaming kernel) -- the merge launch also carries o_proj + residual (B = L = 1).  Against attention + merge + GEMV:
    attention output, residual row and caches BIT-IDENTICAL, ten launches alike, the other output buffer re-armed, the workspace
    left all-ones."""
    from phi_3_vision_mlx_amd.weights import mlx_quantize, q4_repack
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T = cap
    assert ops.attention_decode_can_fuse_oproj(B, L, nh, hd, n_split, T, H, False)
    qkv = g((1, (nh + 2 * nkv) * hd), 245).cuda()
    kc0, vc0 = g((B, nkv, T, hd), 246).cuda(), g((B, nkv, hd, T), 247).cuda()
    wo = (g((H, nh * hd), 248) * 0.05)
    if q4:
        w4, sb = (t.cuda() for t in q4_repack(*mlx_quantize(wo)))
        kw_o = dict(o_proj_w=w4, o_proj_sb=sb)
    else:
        wo = wo.cuda()
        kw_o = dict(o_proj_w=wo)
    x0 = g((1, H), 249).cuda()
    gen = torch.Generator(device="cuda").manual_seed(250)
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda", generator=gen), torch.rand((B, 1, hd // 2), device="cuda", generator=gen)
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=False)
    hp = past - 40 if dev_past else past
    k1, v1, o1, x1 = kc0.clone(), vc0.clone(), torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode(qkv, cos, sin, 1, k1, v1, o1, B, L, nh, nkv, hd, hd ** -0.5, hp, T, ws, n_split, **kw)
    if q4:
        ops.gemv_q4(o1.view(1, H), w4, sb, ops.EPI_RESID_BF16, resid=x1, out=x1)
    else:
        ops.gemv(o1.view(1, H), wo, ops.EPI_RESID_BF16, resid=x1, out=x1)
    for rep in range(10):
        k2, v2, x2 = kc0.clone(), vc0.clone(), x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        ops.attention_decode(qkv, cos, sin, 1, k2, v2, o2, B, L, nh, nkv, hd, hd ** -0.5, hp, T, ws, n_split, **kw, o_proj_x=x2, o_rearm=other, **kw_o)
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + merge + gemv"
        assert torch.equal(k2, k1) and torch.equal(v2, v1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()
    assert not torch.isnan(x1.float()).any() and (x1.float() - x0.float()).abs().max().item() > 0.01


def test_merge_launch_with_fused_oproj_under_grouped_queries(ops):
    """... and with 8 key / value heads for the 32 query heads (the merge launch itself never sees the grouping)."""
    test_merge_launch_with_fused_oproj_is_bit_identical_to_three_launches(ops, 7000, 7168, 112, True, False, nkv=8)
    test_merge_launch_with_fused_oproj_is_bit_identical_to_three_launches(ops, 300, 384, 3, False, True, nkv=8)


@pytest.mark.parametrize("B,L,nh,nkv,hd,K,past,rot,bias,big", [
    (1, 2531, 32, 32, 96, 256, 0, True, False, -1),      # the bench prompt's shape (short K): big tiles + remainder rows, ragged end
    (1, 2531, 32, 32, 96, 256, 0, True, False, 1000000),  # all rows on the 256 x 256 kernel
    (1, 1300, 32, 32, 96, 128, 0, True, False, 0),        # all rows on the 128 x 128 kernel
    (1, 1100, 8, 4, 96, 192, 64, True, False, 512),       # GQA regions of different widths, appended behind 64 cached keys
    (2, 1024, 16, 16, 64, 128, 0, False, True, -1),       # plain head split with bias (CLIP-like), two batch rows, L % 8 == 0
    (3, 680, 16, 16, 64, 128, 8, True, True, 512),
    # short prompts (17 .. 256 rows): the 128 x 64-tile weight-streaming kernel, Q / K tiles in pair order + V tiles with swapped roles
    (1, 128, 32, 32, 96, 256, 0, True, False, -1), (1, 100, 32, 32, 96, 192, 0, True, False, -1), (1, 17, 32, 32, 96, 128, 64, True, False, -1),
    (1, 250, 8, 4, 96, 192, 8, True, False, -1), (2, 64, 16, 16, 64, 128, 0, False, False, -1),
    # round 6: batch rows whose length is not a multiple of 8 (CLIP: 577-token crops) -- the V^T runs fall on even and odd offsets and
    # across row ends
    (17, 577, 16, 16, 64, 1024, 0, False, True, -1), (5, 577, 16, 16, 64, 128, 0, False, True, 0), (3, 1001, 8, 4, 96, 128, 0, True, False, -1),
    (4, 61, 16, 16, 64, 128, 0, False, False, -1), (3, 85, 8, 4, 96, 192, 0, True, False, -1),
])
def test_gemm_qkv_fused(ops, B, L, nh, nkv, hd, K, past, rot, bias, big):
    """p3v_gemm_qkv (qkv projection with head split + rotation + KV append in its epilogue) against the two calls it replaces --
    p3v_gemm then p3v_rope_kv_append -- on the same inputs: rotated / scaled Q, the appended K rows and the appended V^T columns
    BIT-IDENTICAL, everything else in the caches untouched."""
    M, N = B * L, (nh + 2 * nkv) * hd
    a = g((M, K), 400).cuda()
    w = g((N, K), 401, 1.0 / math.sqrt(K)).cuda()
    bq = g((N,), 402, 0.5).cuda() if bias else None
    T = past + L + 5
    Tp = (T + 127) // 128 * 128
    half = hd // 2
    cos = torch.rand((B, T, half), device="cuda") if rot else None
    sin = torch.rand((B, T, half), device="cuda") if rot else None
    qs = 1.3
    k0, v0 = g((B, nkv, Tp, hd), 403).cuda(), g((B, nkv, hd, Tp), 404).cuda()
    pinned = ops.set_tuning("gemm_big_rows", big)
    try:
        qkv = ops.gemm(a, w, ops.EPI_BIAS, bias=bq) if bias else ops.gemm(a, w)
        q1, k1, v1 = torch.empty((B, nh, L, hd), dtype=BF16, device="cuda"), k0.clone(), v0.clone()
        ops.rope_kv_append(qkv, cos, sin, q1, k1, v1, B, L, nh, nkv, hd, past, Tp, True, T, 1, q_scale=qs)
        q2, k2, v2 = torch.full((B, nh, L, hd), 7.0, dtype=BF16, device="cuda"), k0.clone(), v0.clone()
        assert ops.gemm_qkv(a, w, cos, sin, q2, k2, v2, B, L, nh, nkv, hd, past, Tp, True, T, 1, q_scale=qs, bias=bq), "shape not taken"
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("gemm_big_rows", pinned)
    assert torch.equal(q2, q1), f"Q differs ({(q2 != q1).sum().item()} elements)"
    assert torch.equal(k2, k1), f"K cache differs ({(k2 != k1).sum().item()} elements)"
    assert torch.equal(v2, v1), f"V^T cache differs ({(v2 != v1).sum().item()} elements)"
    assert not torch.equal(k1, k0) and not torch.equal(v1, v0)


def test_gemm_qkv_declines_what_it_cannot_fuse(ops):
    """Decode-sized inputs (<= 8 rows) and unaligned append offsets: P3V_ERR_UNSUPPORTED, nothing written."""
    nh, hd, K = 16, 64, 128
    N = 3 * nh * hd
    w = g((N, K), 410).cuda()
    for B, L, past in ((1, 8, 0), (1, 1200, 3), (2, 1001, 5)):
        a = g((B * L, K), 411).cuda()
        Tp = (past + L + 127) // 128 * 128
        q = torch.zeros((B, nh, L, hd), dtype=BF16, device="cuda")
        k, v = torch.zeros((B, nh, Tp, hd), dtype=BF16, device="cuda"), torch.zeros((B, nh, hd, Tp), dtype=BF16, device="cuda")
        assert not ops.gemm_qkv(a, w, None, None, q, k, v, B, L, nh, nh, hd, past, Tp, True)
        torch.cuda.synchronize()
        assert not q.any() and not k.any() and not v.any()


@pytest.mark.parametrize("B,nkv,T,n_tok", [(1, 32, 384, 300), (2, 4, 128, 128), (1, 2, 64, 5)])
def test_kv_quantize_mlx4_is_mx_quantize(ops, B, nkv, T, n_tok):
    """p3v_kv_quantize_mlx4 (the reference's own prompt-cache format, phi.py:528-540) against the MLX group quantiser
    (weights.mlx_quantize: group 32, 4 bits; fp32 arithmetic for the reference's fp32 keys, every primitive of mlx 0.15.0's composite
    rounding to bf16 for its bf16 values) on the same bf16 rows: codes, scales and biases BIT-IDENTICAL for K (token-major) and V
    (stored transposed), the cache rows rewritten with mx.dequantize's values, rows beyond n_tok untouched."""
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    hd = 96
    k = g((B, nkv, T, hd), 300, 1.5)
    v = g((B, nkv, T, hd), 301, 0.7)
    k[0, 0, 1] = 0.0                                        # an all-zero group (scale clamps to 1e-7, every code 0)
    k[0, 0, 2, :32] = torch.linspace(-3, -1, 32).to(BF16)   # all negative: the minimum is the exact edge
    k[0, 0, 3, :32] = torch.linspace(0.5, 2, 32).to(BF16)   # all positive: the maximum is
    kc, vt = k.cuda(), v.transpose(2, 3).contiguous().cuda()
    k4 = torch.zeros((B, nkv, n_tok, 3, 4), dtype=torch.int32, device="cuda")
    v4 = torch.zeros_like(k4)
    ksb = torch.zeros((B, nkv, n_tok, 3, 2), dtype=F32, device="cuda")
    vsb = torch.zeros_like(ksb)
    ops.kv_quantize_mlx4(kc, vt, k4, v4, ksb, vsb, n_tok)
    for name, src, c4, sb, back in (("K", k, k4, ksb, kc), ("V", v, v4, vsb, vt.transpose(2, 3))):
        flat = src[:, :, :n_tok].reshape(B * nkv, n_tok * hd)
        # (the reference's keys reach mx.quantize as fp32 -- the rotation promotes them -- and its values as bf16: scale / bias
        #  come back in the input's dtype and the codes are taken against them)
        pw, ps, pb = mlx_quantize(flat.float() if name == "K" else flat, 32, 4)
        assert torch.equal(c4.cpu().reshape(B * nkv, -1), pw), f"{name}: codes differ from mx.quantize"
        assert torch.equal(sb.cpu()[..., 0].reshape(B * nkv, -1), ps.float()) and torch.equal(sb.cpu()[..., 1].reshape(B * nkv, -1), pb.float()), f"{name}: scale / bias"
        # mx.dequantize: fp32 arrays for the keys (then the cache's bf16), bf16 arrays -- multiply and add each rounding -- for V
        deq = (mlx_dequantize(pw, ps, pb, 32, 4).to(BF16) if name == "K" else mlx_dequantize(pw, ps, pb, 32, 4, dtype=BF16)).reshape(B, nkv, n_tok, hd)
        assert torch.equal(back.cpu()[:, :, :n_tok], deq), f"{name}: rows not rewritten with the dequantised values"
        assert torch.equal(back.cpu()[:, :, n_tok:], src[:, :, n_tok:]), f"{name}: rows beyond n_tok touched"


def test_kv_quantize_mlx4_keys_from_exact_fp32_rotation(ops):
    """With the projection output at hand the keys are quantised as the reference quantises them: from the fp32 result of the rotation
    (phi.py:451 promotes k to fp32 and the cache takes it unrounded, :531), not from the bf16 cache rows.  Codes / scales / biases
    against mx.quantize of the rotation evaluated in fp32 on the host (multiply, multiply, add: the array expression of phi.py:423)."""
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    B, nh, nkv, hd, L, T = 2, 4, 2, 96, 40, 128
    half = hd // 2
    qkv = g((B * L, (nh + 2 * nkv) * hd), 310, 1.2)
    cos, sin = torch.rand((B, 50, half)), torch.rand((B, 50, half))
    x = qkv.view(B, L, nh + 2 * nkv, hd)[:, :, nh:nh + nkv].float()                    # [B, L, nkv, hd] keys before rotation
    k1, k2 = x[..., :half], x[..., half:]
    cs, sn = cos[:, :L, None, :], sin[:, :L, None, :]
    rot = torch.cat([k1 * cs - k2 * sn, k2 * cs + k1 * sn], dim=-1).permute(0, 2, 1, 3).contiguous()   # [B, nkv, L, hd] fp32
    kc = torch.zeros((B, nkv, T, hd), dtype=BF16, device="cuda")
    kc[:, :, :L] = rot.to(BF16).cuda()                                                  # what the rotation kernel left in the cache
    vt = g((B, nkv, hd, T), 311).cuda()
    k4 = torch.zeros((B, nkv, L, 3, 4), dtype=torch.int32, device="cuda")
    v4 = torch.zeros_like(k4)
    ksb = torch.zeros((B, nkv, L, 3, 2), dtype=F32, device="cuda")
    vsb = torch.zeros_like(ksb)
    ops.kv_quantize_mlx4(kc, vt, k4, v4, ksb, vsb, L, qkv=qkv.cuda(), cos_t=cos.cuda(), sin_t=sin.cuda(), nh=nh, past=0, tab_t=50, tab_div=1)
    pw, ps, pb = mlx_quantize(rot.reshape(B * nkv, L * hd), 32, 4)
    assert torch.equal(k4.cpu().reshape(B * nkv, -1), pw), "codes differ from mx.quantize of the fp32 keys"
    assert torch.equal(ksb.cpu()[..., 0].reshape(B * nkv, -1), ps) and torch.equal(ksb.cpu()[..., 1].reshape(B * nkv, -1), pb)
    assert torch.equal(kc.cpu()[:, :, :L], mlx_dequantize(pw, ps, pb, 32, 4).to(BF16).reshape(B, nkv, L, hd))


@pytest.mark.parametrize("seed", range(4))
def test_attention_decode_random_shapes_merge_modes_agree(ops, seed):
    """Fuzz of the decode attention: random (B, L, past, heads, pads) on the 64-key, 128-key and streaming plans; the
    in-launch merge and the merge launch must agree (same partials, different summation order of the splits) and stay
    finite, on ONE workspace that every launch must leave all-sentinel."""
    rng = np.random.default_rng(1234 + seed)
    hd = 96
    ws = ops.attention_ws(4, 16, 8, hd, 64, "cuda")
    for case in range(10):
        B, L, nh = int(rng.integers(1, 4)), int(rng.choice([1, 1, 1, 2, 5, 16])), int(rng.choice([2, 4, 8]))
        past = int(rng.integers(0, 3000))
        plan = rng.choice(["t64", "t128", "stream"])
        if plan == "t128":
            T = (past + L + 3 + 127) // 128 * 128
            n_split = T // 128
        elif plan == "t64":
            T = (past + L + 3 + 63) // 64 * 64
            n_split = T // 64
        else:
            T = (past + L + 3 + 127) // 128 * 128
            n_split = int(rng.integers(1, 5))
        if n_split > 64:
            continue
        pads = rng.integers(0, max(1, past // 2 + 1), size=B).astype(np.int32) if rng.random() < 0.5 else None
        qkv = g((B * L, 3 * nh * hd), 500 + case)
        kc, vc = g((B, nh, T, hd), 600 + case), g((B, nh, T, hd), 700 + case)
        torch.manual_seed(case)
        cos, sin = torch.rand((B, L, hd // 2)).cuda(), torch.rand((B, L, hd // 2)).cuda()
        outs = []
        for fused in (True, False):
            kcc, vcc = kc.cuda(), vc.transpose(2, 3).contiguous().cuda()
            out = torch.full((B, L, nh * hd), float("nan"), dtype=BF16).cuda()
            ops.attention_decode(qkv.cuda(), cos, sin, L, kcc, vcc, out, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split,
                                 pad_len=torch.from_numpy(pads).cuda() if pads is not None else None, merge_in_launch=fused)
            assert torch.isfinite(out.float()).all(), (case, plan, B, L, past, nh, n_split, fused)
            assert (ws.view(torch.int32) == -1).all()
            outs.append(out.float().cpu())
        close(outs[0], outs[1], rtol=2 ** -7, atol=1e-2)


def test_attention_decode_workspace_is_reusable_across_modes_and_shapes(ops, orc):
    """One workspace, used by launches of different shapes and with / without the in-launch merge, in any order: the
    partial records are validated against the 'not written yet' sentinel, so a launch that left anything else behind would
    make a later in-launch merge read stale partials.  Every launch is checked against the first result of its shape."""
    hd, nh = 96, 4
    shapes = [(1, 1, 900, 8), (2, 3, 250, 2), (1, 1, 900, 8), (1, 16, 100, 1), (2, 3, 250, 2)]   # (B, L, past, n_split of 128-key tiles or fewer)
    ws = ops.attention_ws(2, 16, nh, hd, 32, "cuda")                        # big enough for every shape
    seen = {}
    for rep, fused in enumerate([True, False, True, True, False, True]):
        for B, L, past, _ in shapes:
            T = (past + L + 5 + 127) // 128 * 128
            n_split = T // 128
            qkv = g((B * L, 3 * nh * hd), 50 + B + L)
            kc, vc = g((B, nh, T, hd), 60 + B), g((B, nh, T, hd), 70 + L)
            torch.manual_seed(7)
            cos, sin = torch.rand((B, L, hd // 2)).cuda(), torch.rand((B, L, hd // 2)).cuda()
            kcc, vcc = kc.cuda(), vc.transpose(2, 3).contiguous().cuda()
            out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
            ops.attention_decode(qkv.cuda(), cos, sin, L, kcc, vcc, out, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, merge_in_launch=fused)
            assert not torch.isnan(out.float()).any()
            assert (ws.view(torch.int32) == -1).all()
            key = (B, L, past)
            if key in seen:
                close(out, seen[key], rtol=2 ** -7, atol=1e-2)               # merged in the launch == merged by the combine kernel
            else:
                seen[key] = out.float().cpu()


def _dequant(u8, sc, axis):
    return (u8.float() - 128.0) * sc.unsqueeze(axis)


@pytest.mark.parametrize("past,L,n_split,fused", [(200, 1, 3, False), (200, 1, 4, False), (190, 3, 4, False), (62, 5, 4, False),
                                                   (200, 1, 4, True), (62, 5, 4, True),
                                                   (200, 1, 2, False), (200, 1, 2, True), (126, 5, 2, True), (30, 16, 2, False)])
def test_kv_quantize_and_q8_decode(ops, orc, past, L, n_split, fused):
    """int8 KV: (a) quantiser = round(x/s)+128 with s = amax/127 per token, (b) the q8 decode attention equals the
    bf16 oracle attention over the DEQUANTISED cache, (c) the appended rows are stored quantised.
    n_split 3 -> multi-tile single-wave kernel, 4 (= tiles) -> single-tile 4-wave kernel, 2 (= one split per 128 keys) ->
    k_attn_decode128_q8 (raw bytes by LDS-DMA, fp16 conversion at fragment read); (62, 5) / (126, 5) cross a tile boundary;
    fused: the split partials are merged inside the attention launch (ready flags), twice in a row (the flags re-arm)."""
    from phi_3_vision_mlx_amd.config import make_config, rope_scaling_factor
    cfg = make_config()
    B, nh, hd, T = 2, 2, 96, 256
    k, v = g((B, nh, T, hd), 90), g((B, nh, T, hd), 91)
    k8 = torch.full((B, nh, T, hd), 128, dtype=torch.uint8).cuda()
    v8 = torch.full((B, nh, hd, T), 128, dtype=torch.uint8).cuda()
    ksc, vsc = torch.ones((B, nh, T)).cuda(), torch.ones((B, nh, T)).cuda()
    ops.kv_quantize(k.cuda(), v.transpose(2, 3).contiguous().cuda(), k8, v8, ksc, vsc, 0, past)
    for src, q8, sc, ax in ((k, k8.cpu(), ksc.cpu(), -1), (v, v8.cpu().transpose(2, 3), vsc.cpu(), -1)):
        s_ref = src[:, :, :past].float().abs().amax(-1) / 127
        assert torch.allclose(sc[:, :, :past], s_ref, rtol=1e-6)
        q_ref = torch.round(src[:, :, :past].float() / s_ref[..., None]) + 128
        assert (q8[:, :, :past].float() - q_ref).abs().max() <= 1          # ties may round either way
        assert (q8[:, :, past:] == 128).all()
    kd, vd = _dequant(k8.cpu(), ksc.cpu(), -1), _dequant(v8.cpu().transpose(2, 3), vsc.cpu(), -1)
    qkv = g((B * L, 3 * nh * hd), 92)
    inv = 1.0 / (torch.tensor(cfg.rope_scaling["short_factor"], dtype=F32) * (10000.0 ** (torch.arange(0, hd, 2, dtype=F32) / hd)))
    cos, sin = ops.rope_table(torch.arange(T, dtype=F32).repeat(B).cuda(), inv.cuda(), rope_scaling_factor(cfg))
    cos, sin = cos.view(B, T, -1), sin.view(B, T, -1)
    out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    for _ in range(2 if fused else 1):
        out.fill_(float("nan"))
        ops.attention_decode_q8(qkv.cuda(), cos[:, past:], sin[:, past:], T, k8, v8, ksc, vsc, out, B, L, nh, nh, hd, hd ** -0.5, past,
                                T, ws, n_split, merge_in_launch=fused)
    assert (ws.view(torch.int32) == -1).all()                  # every launch leaves the workspace all-sentinel
    cos_ref, sin_ref = orc.su_rope_tables(cfg, T, None)
    x = qkv.view(B, L, 3 * nh, hd).transpose(1, 2)
    cs, sn = cos_ref[:, :, past:past + L], sin_ref[:, :, past:past + L]
    q = orc.rotate_half(x[:, :nh], cs, sn).to(BF16)
    k_new = orc.rotate_half(x[:, nh:2 * nh], cs, sn).to(BF16)
    def qdq(t):                                                 # the step attends over the values it stores
        sc = t.float().abs().amax(-1, keepdim=True) / 127
        return torch.round(t.float() / sc) * sc
    kf = torch.cat([kd[:, :, :past], qdq(k_new)], dim=2)
    vf = torch.cat([vd[:, :, :past], qdq(x[:, 2 * nh:])], dim=2)
    allowed = (torch.arange(past + L)[None, :] <= (past + torch.arange(L))[:, None])[None, None].expand(B, 1, L, past + L)
    ref = _attn_ref(orc, q, kf, vf, hd ** -0.5, allowed).transpose(1, 2).reshape(B, L, nh * hd)
    close(out, ref, rtol=2 ** -6, atol=2e-2)
    s_new = k_new.float().abs().amax(-1) / 127
    assert torch.allclose(ksc.cpu()[:, :, past:past + L], s_new, rtol=1e-6)
    assert (k8.cpu()[:, :, past:past + L].float() - (torch.round(k_new.float() / s_new[..., None]) + 128)).abs().max() <= 1
    assert (v8.cpu()[:, :, :, past].float() - (torch.round(x[:, 2 * nh:, 0].float() / (x[:, 2 * nh:, 0].float().abs().amax(-1, keepdim=True) / 127)) + 128)).abs().max() <= 1


@pytest.mark.parametrize("past,cap,dev_past", [(2531, 2688, True), (300, 1664, False)])
def test_attention_decode_with_fused_4bit_oproj_is_bit_identical_to_two_launches(ops, past, cap, dev_past):
    """k_attn_decode128_o4: the fused launch on MLX 4-bit group-64 o_proj weights (the reference's quantize_model format, device
    layout of weights.q4_repack) against p3v_attention_decode + p3v_gemv_q4(P3V_EPI_RESID_BF16): bit-identical, ten times."""
    from phi_3_vision_mlx_amd.weights import mlx_quantize, q4_repack
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T, n_split = cap, cap // 128
    qkv = g((1, 3 * nh * hd), 165).cuda()
    kc0, vc0 = g((B, nh, T, hd), 166).cuda(), g((B, nh, hd, T), 167).cuda()
    w4, sb = (t.cuda() for t in q4_repack(*mlx_quantize(g((H, nh * hd), 168) * 0.05)))
    x0 = g((1, H), 169).cuda()
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda"), torch.rand((B, 1, hd // 2), device="cuda")
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=True)
    hp = past - 40 if dev_past else past
    k1, v1, o1, x1 = kc0.clone(), vc0.clone(), torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode(qkv, cos, sin, 1, k1, v1, o1, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw)
    ops.gemv_q4(o1.view(1, H), w4, sb, ops.EPI_RESID_BF16, resid=x1, out=x1)
    for rep in range(10):
        k2, v2, x2 = kc0.clone(), vc0.clone(), x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        ops.attention_decode(qkv, cos, sin, 1, k2, v2, o2, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw,
                             o_proj_w=w4, o_proj_sb=sb, o_proj_x=x2, o_rearm=other)
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + gemv_q4"
        assert torch.equal(k2, k1) and torch.equal(v2, v1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()
    assert not torch.isnan(x1.float()).any() and (x1.float() - x0.float()).abs().max().item() > 0.01


@pytest.mark.parametrize("past,cap,dev_past", [(2531, 2688, True), (300, 1664, False), (2559, 2688, True)])
def test_q8_attention_decode_with_fused_fp8_oproj_is_bit_identical_to_two_launches(ops, past, cap, dev_past):
    """k_attn_decode128_q8<true> (config 5): int8-KV attention + e4m3 o_proj + residual in ONE launch against
    p3v_attention_decode_q8 then p3v_gemv_fp8(P3V_EPI_RESID_BF16): residual row, attention output, appended bytes and scales
    bit-identical, ten launches in a row the same, the other output buffer re-armed, the workspace left all-ones."""
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T, n_split = cap, cap // 128
    assert ops.attention_decode_q8_can_fuse_oproj(B, L, nh, hd, n_split, T, H, True)
    assert not ops.attention_decode_q8_can_fuse_oproj(2, L, nh, hd, n_split, T, H, True)
    qkv = g((1, 3 * nh * hd), 155).cuda()
    k, v = g((B, nh, T, hd), 156).cuda(), g((B, nh, hd, T), 157).cuda()
    k8_0 = torch.full((B, nh, T, hd), 128, dtype=torch.uint8).cuda()
    v8_0 = torch.full((B, nh, hd, T), 128, dtype=torch.uint8).cuda()
    ks0, vs0 = torch.ones((B, nh, T)).cuda(), torch.ones((B, nh, T)).cuda()
    ops.kv_quantize(k, v, k8_0, v8_0, ks0, vs0, 0, past)
    w8, wsc = ops.quantize_fp8_rows((g((H, nh * hd), 158) * 0.05).cuda())
    x0 = g((1, H), 159).cuda()
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda"), torch.rand((B, 1, hd // 2), device="cuda")
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=True)
    def fresh():
        return k8_0.clone(), v8_0.clone(), ks0.clone(), vs0.clone()
    k1, v1, ks1, vs1 = fresh()
    o1, x1 = torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode_q8(qkv, cos, sin, 1, k1, v1, ks1, vs1, o1, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, **kw)
    ops.gemv_fp8(o1.view(1, H), w8, wsc, ops.EPI_RESID_BF16, resid=x1, out=x1)
    for rep in range(10):
        k2, v2, ks2, vs2 = fresh()
        x2 = x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        ops.attention_decode_q8(qkv, cos, sin, 1, k2, v2, ks2, vs2, o2, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, **kw,
                                o_proj_w8=w8, o_proj_scale=wsc, o_proj_x=x2, o_rearm=other)
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + gemv_fp8"
        assert torch.equal(k2, k1) and torch.equal(v2, v1) and torch.equal(ks2, ks1) and torch.equal(vs2, vs1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()
    assert not torch.isnan(x1.float()).any() and (x1.float() - x0.float()).abs().max().item() > 0.01


@pytest.mark.parametrize("past,cap,n_split,dev_past", [(7000, 7168, 56, False), (9000, 9216, 40, True), (600, 640, 5, False), (600, 640, 10, True)])
def test_q8_merge_launch_with_fused_fp8_oproj_is_bit_identical_to_three_launches(ops, past, cap, n_split, dev_past):
    """k_attn_combine_o<e4m3> (round 6, config 5 at long contexts): the int8-KV plans whose partials are merged by a launch of their own
    (128-key tiles without the in-launch merge, the multi-tile single-wave kernel, the 64-key one-tile kernel) -- the merge launch also
    carries the e4m3 o_proj + residual.  Against attention + merge + p3v_gemv_fp8: everything BIT-IDENTICAL, ten launches alike."""
    B, L, nh, hd, H = 1, 1, 32, 96, 3072
    T = cap
    assert ops.attention_decode_q8_can_fuse_oproj(B, L, nh, hd, n_split, T, H, False)
    assert not ops.attention_decode_q8_can_fuse_oproj(2, L, nh, hd, n_split, T, H, False)
    qkv = g((1, 3 * nh * hd), 355).cuda()
    k, v = g((B, nh, T, hd), 356).cuda(), g((B, nh, hd, T), 357).cuda()
    k8_0 = torch.full((B, nh, T, hd), 128, dtype=torch.uint8).cuda()
    v8_0 = torch.full((B, nh, hd, T), 128, dtype=torch.uint8).cuda()
    ks0, vs0 = torch.ones((B, nh, T)).cuda(), torch.ones((B, nh, T)).cuda()
    ops.kv_quantize(k, v, k8_0, v8_0, ks0, vs0, 0, past)
    w8, wsc = ops.quantize_fp8_rows((g((H, nh * hd), 358) * 0.05).cuda())
    x0 = g((1, H), 359).cuda()
    gen = torch.Generator(device="cuda").manual_seed(360)
    cos, sin = torch.rand((B, 1, hd // 2), device="cuda", generator=gen), torch.rand((B, 1, hd // 2), device="cuda", generator=gen)
    d_past = torch.tensor([past], dtype=torch.int32).cuda()
    ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
    kw = dict(d_past=d_past if dev_past else None, merge_in_launch=False)
    def fresh():
        return k8_0.clone(), v8_0.clone(), ks0.clone(), vs0.clone()
    k1, v1, ks1, vs1 = fresh()
    o1, x1 = torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
    ops.attention_decode_q8(qkv, cos, sin, 1, k1, v1, ks1, vs1, o1, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, **kw)
    ops.gemv_fp8(o1.view(1, H), w8, wsc, ops.EPI_RESID_BF16, resid=x1, out=x1)
    for rep in range(10):
        k2, v2, ks2, vs2 = fresh()
        x2 = x0.clone()
        o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
        other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
        ops.attention_decode_q8(qkv, cos, sin, 1, k2, v2, ks2, vs2, o2, B, L, nh, nh, hd, hd ** -0.5, past, T, ws, n_split, **kw,
                                o_proj_w8=w8, o_proj_scale=wsc, o_proj_x=x2, o_rearm=other)
        assert torch.equal(o2.view(torch.int16), o1.view(torch.int16)), f"rep {rep}: attention output differs"
        assert torch.equal(x2.view(torch.int16), x1.view(torch.int16)), f"rep {rep}: residual row differs from attention + merge + gemv_fp8"
        assert torch.equal(k2, k1) and torch.equal(v2, v1) and torch.equal(ks2, ks1) and torch.equal(vs2, vs1)
        assert (other.view(torch.int16) == -1).all() and (ws.view(torch.int32) == -1).all()
    assert not torch.isnan(x1.float()).any() and (x1.float() - x0.float()).abs().max().item() > 0.01


def test_attention_beam_view(ops, orc):
    """n_beam: keys [0,past) come from cache row b//n_beam, new keys from a scratch (phi.py:523-527)."""
    Bc, nb, L, past, nh, hd = 2, 3, 4, 50, 2, 96
    B, T = Bc * nb, 64
    q = g((B, nh, L, hd), 50)
    kc, vc = g((Bc, nh, T, hd), 51), g((Bc, nh, T, hd), 52)
    kn, vn = g((B, nh, L, hd), 53), g((B, nh, L, hd), 54)
    out = torch.empty((B, L, nh * hd), dtype=BF16).cuda()
    ws = ops.attention_ws(B, L, nh, hd, 2, "cuda")
    Lp = 8
    knp = torch.zeros((B, nh, Lp, hd), dtype=BF16)
    vnp = torch.zeros((B, nh, hd, Lp), dtype=BF16)
    knp[:, :, :L], vnp[:, :, :, :L] = kn, vn.transpose(2, 3)
    ops.attention(q.cuda(), out, B, L, nh, nh, hd, hd ** -0.5, True, k_new=knp.cuda(), v_new=vnp.cuda(), new_t=Lp, past=past,
                  k_past=kc.cuda(), v_past=vc.transpose(2, 3).contiguous().cuda(), past_t=T, past_div=nb, ws=ws, n_split=2)
    kf = torch.cat([kc[:, :, :past].repeat_interleave(nb, dim=0), kn], dim=2)
    vf = torch.cat([vc[:, :, :past].repeat_interleave(nb, dim=0), vn], dim=2)
    allowed = (torch.arange(past + L)[None, :] <= (past + torch.arange(L))[:, None])[None, None].expand(B, 1, L, past + L)
    ref = _attn_ref(orc, q, kf, vf, hd ** -0.5, allowed).transpose(1, 2).reshape(B, L, nh * hd)
    close(out, ref, rtol=2 ** -6, atol=2e-2)


def test_im2col_cls_hd_merge(ops, orc):
    n, D = 3, 128
    pix = g((n, 3, 336, 336), 60, 1.0, F32)
    pat = ops.im2col_patches(pix.cuda(), 14, 640)
    ref = torch.nn.functional.unfold(pix, 14, stride=14).transpose(1, 2).reshape(n * 576, 588)
    assert torch.equal(pat[:, :588].cpu(), ref.to(BF16)) and pat[:, 588:].abs().sum().item() == 0
    # HD merge vs the oracle's reshape/transpose/concat (phi.py:403-407)
    from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
    cfg = make_config(tiny_config_dict())
    C_ = cfg.img_processor["image_dim_out"]
    for h, w in [(1, 1), (2, 3), (4, 4), (3, 4)]:
        nc = h * w + 1
        feats = g((nc, 577, C_), 61, 1.0, F32)
        sub, glb = g((1, 1, 1, 4 * C_), 62), g((1, 1, 4 * C_), 63)
        out = ops.hd_merge(feats.cuda(), sub.cuda(), glb.cuda(), h, w, 24, C_)
        o = orc.OraclePhi3V.__new__(orc.OraclePhi3V)
        f = feats[:, 1:]                                            # drop CLS
        def rc(img, shape, tile):
            t = img.reshape(shape).permute(0, 1, 3, 2, 4, 5).reshape(tile)
            return torch.cat([t, sub.float().expand(1, tile[1], 1, -1)], dim=2).reshape(1, -1, 4 * C_)
        g_ = rc(f[:1], (1, 12, 2, 12, 2, C_), (1, 12, 12, 4 * C_))
        s_ = rc(f[1:], (h * w, 12, 2, 12, 2, C_), (1, h * 12, w * 12, 4 * C_))
        ref = torch.cat([s_, glb.float(), g_], dim=1)[0]
        assert out.shape[0] == (h * w + 1) * 144 + 1 + (h + 1) * 12
        assert torch.equal(out.cpu(), ref.to(BF16))


def test_argmax_logsoftmax_topk(ops, orc):
    x = g((5, 32064), 70, 2.0)
    x[1, 100] = x[1, 31000] = x[1].float().max() + 1            # tie -> first index
    x[2, :] = -1.5
    xc = x.cuda()
    assert ops.argmax(xc).cpu().tolist() == torch.argmax(x.float(), dim=-1).tolist()
    assert ops.argmax(xc).cpu().tolist()[1] == 100 and ops.argmax(xc).cpu().tolist()[2] == 0
    close(ops.log_softmax(xc), orc.log_softmax(x), rtol=2 ** -7, atol=1e-2)
    tk = ops.topk(xc, 3).cpu()
    assert tk.tolist() == orc.top3_candidates(x, 3).tolist()


def test_argmax_reports_nan_rows(ops):
    """A NaN anywhere in a row makes its arg-max -1 (the host loops raise on it), other rows are unaffected."""
    from phi_3_vision_mlx_amd.api import _rows
    x = g((3, 32064), 77)
    x[1, 20000] = float("nan")
    got = ops.argmax(x.cuda())
    assert got.tolist() == [int(x[0].float().argmax()), -1, int(x[2].float().argmax())]
    with pytest.raises(RuntimeError):
        _rows(got)


def test_graph_replay(ops):
    """hipGraph capture of a launch sequence replays with updated inputs."""
    x, w = g((4, 192), 80).cuda(), (g((192,), 81, 0.1) + 1).cuda()
    y = torch.empty_like(x)
    ops.rmsnorm(x, w, 1e-5, out=y)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = ops.Graph()
        gr.begin()
        ops.rmsnorm(x, w, 1e-5, out=y)
        gr.end()
        x.copy_(g((4, 192), 82))
        gr.launch()
    s.synchronize()
    import phi3v_oracle as orc_
    close(y, orc_.rms_norm(x.cpu(), w.cpu(), 1e-5), rtol=2 ** -6, atol=1e-3)


@pytest.mark.parametrize("B", [1, 5])
def test_step_begin_end_match_separate_kernels(ops, B):
    """The fused head/tail of a replayed greedy step = embed_gather + stage_rope / argmax + store_token + 2 x add_i32."""
    V, H, T, half, steps = 32064, 3072, 40, 48, 6
    table = g((V, H), 70).cuda()
    cos, sin = torch.rand((B, T, half), dtype=F32).cuda(), torch.rand((B, T, half), dtype=F32).cuda()
    tok = torch.tensor([(7 * b + 3) % V for b in range(B)], dtype=torch.int32).cuda()
    d_past = torch.tensor([11], dtype=torch.int32).cuda()
    x = torch.empty((B, H), dtype=BF16).cuda()
    c_o, s_o = torch.empty((B, 1, half), dtype=F32).cuda(), torch.empty((B, 1, half), dtype=F32).cuda()
    ops.step_begin(tok, table, x, cos, sin, d_past, c_o, s_o)
    assert torch.equal(x, table[tok.long()])
    assert torch.equal(c_o[:, 0], cos[:, 11]) and torch.equal(s_o[:, 0], sin[:, 11])
    hist = torch.zeros((B, steps), dtype=torch.int32).cuda()
    d_step, ticket = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    nxt = torch.zeros(B, dtype=torch.int32).cuda()
    want = []
    for s in range(steps + 1):                                  # one step past the history capacity: must not write
        lg = g((B, V), 80 + s).cuda()
        lg[:, 5] = lg[:, 17000] = lg.float().max() + 1           # tie: the first maximum wins
        ops.step_end(lg, nxt, tok, hist, d_step, d_past, ticket)
        ref = ops.argmax(lg)
        assert torch.equal(nxt, ref) and torch.equal(tok, ref) and (ref == 5).all()
        want.append(ref.clone())
        assert d_step.item() == s + 1 and d_past.item() == 12 + s and ticket.item() == 0
    assert torch.equal(hist, torch.stack(want[:steps], dim=1))


@pytest.mark.parametrize("K", [3072, 8192])
def test_gemv_step_folds_are_the_separate_launches(ops, K, B=1):
    """p3v_gemv_step (round 6): the replayed step's first projection with step_begin in its prologue, and its last with step_end in
    its epilogue, against the launches they replace -- projection outputs, residual rows, staged rotation rows, arg-max tokens
    (ties: first maximum; NaN rows: -1), history and both counters BIT-IDENTICAL over several steps, the ticket left at zero;
    shapes the fold does not take answer False and launch nothing."""
    V, T, half, steps = 32064, 40, 48, 5
    N1 = 1024                                                     # the "qkv" of this test
    table = g((V, K), 70).cuda()
    cos, sin = torch.rand((B, T, half), dtype=F32).cuda(), torch.rand((B, T, half), dtype=F32).cuda()
    w1, nw1 = (g((N1, K), 71) * 0.05).cuda(), (1 + 0.1 * g((K,), 72)).cuda()
    tok = torch.tensor([V + 5 if K == 8192 else 7004], dtype=torch.int32).cuda()   # (out of range: clamped, as p3v_step_begin clamps)
    d_past = torch.tensor([11], dtype=torch.int32).cuda()
    x_a, x_b = torch.empty((B, K), dtype=BF16).cuda(), torch.full((B, K), 7.0, dtype=BF16).cuda()
    ca, sa = torch.empty((B, 1, half), dtype=F32).cuda(), torch.empty((B, 1, half), dtype=F32).cuda()
    cb, sb = torch.zeros_like(ca), torch.zeros_like(sa)
    ops.step_begin(tok, table, x_a, cos, sin, d_past, ca, sa)
    out_a = ops.gemv(x_a, w1, norm_w=nw1, norm_eps=1e-5)
    out_b = torch.full((B, N1), float("nan"), dtype=BF16).cuda()
    assert ops.gemv_step_begin(tok, table, x_b, cos, sin, d_past, cb, sb, w1, nw1, 1e-5, out_b)
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)) and torch.equal(x_a, x_b)
    assert torch.equal(ca, cb) and torch.equal(sa, sb) and torch.equal(cb[:, 0], cos[:, 11])
    # the tail: final norm + lm_head + arg-max + bookkeeping
    wl, nwl = (g((V, K), 73) * 0.05).cuda(), (1 + 0.1 * g((K,), 74)).cuda()
    amax_ws = torch.zeros((ops.L.GEMV_STEP_WS_BYTES // 4,), dtype=F32).cuda()      # (candidate records + arrival counters: zero once)
    hist_a, hist_b = torch.zeros((B, steps), dtype=torch.int32).cuda(), torch.zeros((B, steps), dtype=torch.int32).cuda()
    st_a, st_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    pa, pb = d_past.clone(), d_past.clone()
    tk_a, tk_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    nx_a, nx_b, to_a, to_b = (torch.zeros(B, dtype=torch.int32).cuda() for _ in range(4))
    for s in range(steps + 1):                                    # one step past the history capacity: must not write
        x = g((B, K), 90 + s).cuda()
        if s == 2:                                                # a tie between two vocabulary rows: the first one wins
            wl[20000] = wl[300]
        if s == 3:                                                # a poisoned row reports -1, the others their arg-max
            x[0, 5] = float("nan")
        lg_a = ops.gemv(x, wl, norm_w=nwl, norm_eps=1e-5)
        ops.step_end(lg_a, nx_a, to_a, hist_a, st_a, pa, tk_a)
        lg_b = torch.empty((B, V), dtype=BF16).cuda()
        assert ops.gemv_step_end(x, wl, nwl, 1e-5, lg_b, nx_b, to_b, hist_b, st_b, pb, tk_b, amax_ws)
        assert torch.equal(lg_a.view(torch.int16), lg_b.view(torch.int16))
        assert torch.equal(nx_a, nx_b) and torch.equal(to_a, to_b), (s, nx_a.tolist(), nx_b.tolist())
        assert st_b.item() == s + 1 and pb.item() == 12 + s and tk_b.item() == 0 and st_a.item() == st_b.item() and pa.item() == pb.item()
        if s == 3:
            assert nx_b[0].item() == -1
    assert torch.equal(hist_a, hist_b)
    # not on the fold's path: more than one row (p3v_gemv runs another kernel there), a hidden size the streaming kernel does not take
    assert not ops.gemv_step_end(g((2, K), 1).cuda(), wl, nwl, 1e-5, torch.empty((2, V), dtype=BF16).cuda(), nx_b, to_b, hist_b, st_b, pb, tk_b, amax_ws)
    small = g((64, 192), 2).cuda()
    assert not ops.gemv_step_begin(tok[:1], g((V, 192), 3).cuda(), torch.empty((1, 192), dtype=BF16).cuda(), cos, sin, d_past, cb, sb, small,
                                   torch.ones(192, dtype=BF16).cuda(), 1e-5, torch.empty((1, 64), dtype=BF16).cuda())
    assert st_b.item() == steps + 1                               # (nothing was launched)


@pytest.mark.parametrize("M,K,N,r", [(1, 3072, 9216, 1), (5, 192, 576, 8), (300, 256, 512, 20)])
def test_lora_down_up(ops, M, K, N, r):
    """p3v_lora_down / p3v_lora_up vs LoRALinear.__call__ (phi.py:129-133) in fp32, all three fused epilogues."""
    from phi_3_vision_mlx_amd.ops import EPI_NONE, EPI_RESID_BF16, EPI_SILU_MUL
    x, y, res = g((M, K), 100), g((M, N), 101), g((M, N), 102)
    a = torch.randn((K, r), generator=torch.Generator().manual_seed(103)) * K ** -0.5
    b = torch.randn((r, N), generator=torch.Generator().manual_seed(104)) * 0.5
    scale = 1.7
    t = ops.lora_down(x.cuda(), a.cuda())
    t_ref = x.float() @ a
    assert torch.allclose(t.cpu(), t_ref, rtol=1e-4, atol=1e-4)
    v = (y.float() + scale * (t_ref @ b)).to(BF16)
    close(ops.lora_up(y.cuda(), t, b.cuda(), scale, EPI_NONE), v, rtol=2 ** -7, atol=1e-2)
    close(ops.lora_up(y.cuda(), t, b.cuda(), scale, EPI_RESID_BF16, resid=res.cuda()), (res.float() + v.float()).to(BF16), rtol=2 ** -7, atol=2e-2)
    gate, up = v[:, :N // 2], v[:, N // 2:]
    act = gate * torch.sigmoid(gate)
    close(ops.lora_up(y.cuda(), t, b.cuda(), scale, EPI_SILU_MUL), act * up, rtol=2 ** -6, atol=2e-2)



@pytest.mark.parametrize("w,h,kind,seed", [(336, 336, "noise", 0), (640, 480, "gradient", 1), (500, 1000, "noise", 2), (1600, 1200, "noise", 3),
                                          (90, 61, "noise", 4)])
def test_device_preprocessing_is_bit_identical_to_host(w, h, kind, seed):
    """p3v_resample_u8 x 2 + p3v_hd_preprocess == the host image processor (itself sha256-pinned to the reference):
    square, landscape, portrait (transposed path), down-scaled and tiny inputs."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import numpy as np
    from golden_inputs import make_image
    from phi_3_vision_mlx_amd.processor import Phi3VImageProcessor
    p = Phi3VImageProcessor()
    img = make_image(w, h, kind, seed)
    host = p([img], dtype=np.float32)
    dev = p.device_call([img], "cuda:0")
    assert dev["image_sizes"] == host["image_sizes"] and dev["num_img_tokens"] == host["num_img_tokens"]
    got = dev["pixel_values"].cpu().numpy()
    assert got.dtype == np.float32 and got.shape == host["pixel_values"].shape
    assert np.array_equal(got.view(np.uint32), host["pixel_values"].view(np.uint32))        # bit patterns, signed zeros included


@pytest.mark.parametrize("N,K,epi", [(3072, 3072, "resid"), (9216, 3072, "none"), (8192, 3072, "silu"), (3072, 8192, "resid"), (512, 8192, "f32")])
@pytest.mark.parametrize("norm", [False, True])
def test_gemv_q4_and_dequant(ops, N, K, epi, norm):
    """4-bit group-64 affine weights (the reference's nn.quantize(model, 64, 4)): p3v_dequant_q4 == scale*q+bias exactly
    (one bf16 rounding), p3v_gemv_q4 == the fp32 product with the DEQUANTISED weights (mx.quantized_matmul semantics)."""
    from phi_3_vision_mlx_amd.ops import EPI_F32, EPI_NONE, EPI_RESID_BF16, EPI_SILU_MUL
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize, q4_repack
    rows = 2 * N if epi == "silu" else N
    w = g((rows, K), 400, 0.03)
    packed, sc, bi = mlx_quantize(w)
    wd = mlx_dequantize(packed, sc, bi)                                   # fp32, exact scale*q+bias
    w4, sb = (t.cuda() for t in q4_repack(packed, sc, bi))
    assert torch.equal(ops.dequant_q4(w4, sb).cpu(), wd.to(BF16))
    x = g((1, K), 401)
    nw = (g((K,), 402) * 0.1 + 1) if norm else None
    xin = x.float()
    if norm:
        xin = ((xin * torch.rsqrt(xin.pow(2).mean(-1, keepdim=True) + 1e-5)).to(BF16) * nw).float()      # mx.fast.rms_norm: two roundings
    y = xin @ wd.t()
    res = g((1, N), 403)
    e = {"none": EPI_NONE, "resid": EPI_RESID_BF16, "silu": EPI_SILU_MUL, "f32": EPI_F32}[epi]
    got = ops.gemv_q4(x.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None, norm_w=nw.cuda() if norm else None, norm_eps=1e-5)
    if epi == "resid":
        ref = (res.float() + y.to(BF16).float()).to(BF16)
    elif epi == "silu":
        gt, up = y[:, :N].to(BF16), y[:, N:].to(BF16)
        ref = (gt * torch.sigmoid(gt)) * up
    else:
        ref = y if epi == "f32" else y.to(BF16)
    close(got, ref, rtol=2 ** -6, atol=3e-2)


@pytest.mark.parametrize("M", [2, 5, 8, 15, 16])
@pytest.mark.parametrize("N,K,epi", [(9216, 3072, "none"), (3072, 3072, "resid"), (8192, 3072, "silu"), (3072, 8192, "resid"), (32064, 3072, "none")])
def test_gemv_q4_rows_2_to_16(ops, N, K, epi, M):
    """2 .. 16 rows straight on the 4-bit weights (k_gemm_rows_q4, round 6): == the fp32 product with the dequantised weights
    (mx.quantized_matmul's semantics) on the decoder's own shapes -- the dequantisation in registers keeps 11 significant bits of
    scale * q + bias (fp16), x is converted to fp16 exactly; against the dequantise-then-bf16-GEMM path it replaces; three launches
    bit-identical; rows beyond M untouched; a fused RMSNorm is declined (the model normalises first)."""
    from phi_3_vision_mlx_amd.ops import EPI_NONE, EPI_RESID_BF16, EPI_SILU_MUL
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize, q4_repack
    rows = 2 * N if epi == "silu" else N
    w = g((rows, K), 410, 0.03)
    packed, sc, bi = mlx_quantize(w)
    wd = mlx_dequantize(packed, sc, bi)
    w4, sb = (t.cuda() for t in q4_repack(packed, sc, bi))
    x = g((M, K), 411 + M)
    res = g((M, N), 413)
    e = {"none": EPI_NONE, "resid": EPI_RESID_BF16, "silu": EPI_SILU_MUL}[epi]
    buf = torch.full((M + 2, N), 7.0, dtype=BF16, device="cuda")
    got = ops.gemv_q4(x.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None, out=buf[:M])
    assert bool((buf[M:] == 7.0).all())
    for _ in range(2):
        assert torch.equal(ops.gemv_q4(x.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None), got)
    y = x.double() @ wd.double().t()
    if epi == "resid":
        ref = (res.double() + y.to(BF16).double()).to(BF16)
    elif epi == "silu":
        gt, up = y[:, :N].to(BF16).double(), y[:, N:].to(BF16).double()
        ref = ((gt * torch.sigmoid(gt)).to(BF16).double() * up).to(BF16)
    else:
        ref = y.to(BF16)
    def near(a, b):
        # SiLU * up: when the bf16 rounding of a gate or an up value falls on the other side of a tie in two correct computations the
        # product moves by a whole ulp of one factor times the other factor: a handful of the 10^5 outputs may sit up to twice the
        # bound away (the M = 1 kernel's test draws 8192 outputs and meets none)
        if epi != "silu":
            return close(a, b, rtol=2 ** -6, atol=3e-2)
        a, b = a.float().cpu(), b.float().cpu()
        err, tol = (a - b).abs(), 3e-2 + 2 ** -6 * b.abs()
        assert (err <= 2 * tol).all() and (err > tol).float().mean().item() < 1e-4, (err / tol).max().item()
    near(got, ref)
    wb = ops.dequant_q4(w4, sb)                                          # the path it replaces: a bf16 scratch + the bf16 kernels
    other = ops.gemm(x.cuda(), wb, e, resid=res.cuda() if epi == "resid" else None) if M > 8 else \
        ops.gemv(x.cuda(), wb, e, resid=res.cuda() if epi == "resid" else None)
    near(got, other)
    nw = (1 + 0.1 * g((K,), 415)).cuda()
    if M > 8:                                                      # (no fused RMSNorm at 9 .. 16 rows: the model normalises first)
        with pytest.raises(RuntimeError):
            ops.gemv_q4(x.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None, norm_w=nw, norm_eps=1e-5)
        return
    # 2 .. 8 rows (k_gemv8_q4): the input RMSNorm rides along -- the value p3v_rmsnorm writes (bf16, exact in fp16), so the fused call
    # equals the two calls up to the summation order of the squares (a last-bit difference of r on a handful of elements)
    ops.set_tuning("gemv_q4_rows8", 0)                             # k_gemm_rows_q4 (also what 9 .. 16 rows run): same product, other order
    try:
        near(ops.gemv_q4(x.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None), ref)
    finally:
        ops.set_tuning("gemv_q4_rows8", 1)
    xs = (x * 3.0).to(BF16)
    h = ops.rmsnorm(xs.cuda(), nw, 1e-5)
    two = ops.gemv_q4(h, w4, sb, e, resid=res.cuda() if epi == "resid" else None)
    one = ops.gemv_q4(xs.cuda(), w4, sb, e, resid=res.cuda() if epi == "resid" else None, norm_w=nw, norm_eps=1e-5)
    near(one, two)
    assert (one.float() != two.float()).float().mean().item() < 0.02


# ----------------------------------------------------------------------------- W8A8 on the fp8 matrix cores (config 5 prefill)
def _e4m3(x):
    return x.to(torch.float8_e4m3fn)


@pytest.mark.parametrize("rows,K,norm", [(1, 3072, False), (300, 3072, True), (37, 8192, False), (2531, 3072, True)])
def test_quant_fp8_rows(ops, rows, K, norm):
    """Activation quantiser: one scale s per token row = max|h| / 448, codes = e4m3(h * (1 / s)) (RNE), h = x or RMSNorm(x)."""
    x = g((rows, K), 30, 2.0)
    x[0, :] = 0 if rows > 1 else x[0, :]                        # an all-zero row keeps scale 1
    w = (g((K,), 31, 0.1) + 1) if norm else None
    q, s = ops.quant_fp8_rows(x.cuda(), None if w is None else w.cuda(), 1e-5)
    h = x.float()
    if norm:
        h = ((h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + 1e-5)).to(BF16) * w).float()      # mx.fast.rms_norm: two roundings
    amax = h.abs().amax(-1)
    s_ref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q_ref = _e4m3(h * (1.0 / s_ref)[:, None]).view(torch.uint8)
    if not norm:
        assert torch.equal(s.cpu(), s_ref) and torch.equal(q.cpu(), q_ref)
    else:                                                       # rsqrt differs by an ulp between the two sides: a few bf16 flips
        assert torch.allclose(s.cpu(), s_ref, rtol=2 ** -7)
        same = (q.cpu() == q_ref).float().mean().item()
        assert same > 0.995, same
        dq = q.cpu().view(torch.float8_e4m3fn).float() * s.cpu()[:, None]
        assert (dq - h).abs().max().item() <= 2 ** -3 * h.abs().amax().item()


@pytest.mark.parametrize("narrow", [None, "0", "1"])
@pytest.mark.parametrize("epi,M,N,K", [("none", 300, 512, 384), ("none", 2531, 9216, 3072), ("resid", 2531, 3072, 8192),
                                       ("silu", 700, 1024, 3072), ("resid", 17, 256, 128), ("none", 513, 384, 256)])
def test_gemm_fp8(ops, request, epi, M, N, K, narrow):
    """out = epilogue(sa[m] sw[n] sum_k a8[m,k] w8[n,k]) on v_mfma_scale_f32_16x16x128_f8f6f4 vs the same products in fp32
    on the CPU (e4m3 x e4m3 products are exact in fp32; only the summation order differs).  narrow: the launcher's own
    choice of tile width / 256-wide tiles / 128-wide tiles."""
    if narrow is not None:
        if epi == "silu" and narrow == "1":
            pytest.skip("SiLU*up runs on wide tiles only")
        old = ops.set_tuning("gemm_f8_narrow", int(narrow))
        request.addfinalizer(lambda: ops.set_tuning("gemm_f8_narrow", old))
    EPI = {"none": ops.EPI_NONE, "resid": ops.EPI_RESID_BF16, "silu": ops.EPI_SILU_MUL}[epi]
    n_rows = 2 * N if epi == "silu" else N
    a8 = _e4m3(g((M, K), 40, 1.0, F32) * 3).view(torch.uint8)          # asymmetric random codes incl. denormals / zeros
    w8 = _e4m3(g((n_rows, K), 41, 1.0, F32) * 3).view(torch.uint8)
    sa = torch.rand((M,), generator=torch.Generator().manual_seed(42)) * 0.02 + 0.002
    sw = torch.rand((n_rows,), generator=torch.Generator().manual_seed(43)) * 0.01 + 0.001
    resid = g((M, N), 44, 1.0) if epi == "resid" else None
    out = ops.gemm_fp8(a8.cuda(), sa.cuda(), w8.cuda(), sw.cuda(), EPI, resid=None if resid is None else resid.cuda())
    af = a8.view(torch.float8_e4m3fn).float() * sa[:, None]
    wf = w8.view(torch.float8_e4m3fn).float() * sw[:, None]
    y = (af.double() @ wf.double().t()).float()
    if epi == "resid":
        ref = (resid.float() + y.to(BF16).float()).to(BF16)
    elif epi == "silu":
        gte, up = y[:, :N].to(BF16), y[:, N:].to(BF16)
        ref = ((gte * torch.sigmoid(gte)) * up).to(BF16)
    else:
        ref = y.to(BF16)
    assert out.shape == ref.shape
    close(out, ref, rtol=2 ** -6, atol=2e-2 * ref.float().abs().max().item() / 16)


def test_gemm_fp8_asymmetric_identity(ops):
    """A = I (scaled), asymmetric W: catches a transposed / permuted C write and a wrong lane -> k mapping."""
    K = N = 256
    eye = torch.eye(K)
    a8 = _e4m3(eye).view(torch.uint8)
    wv = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 17 - 8).float()      # exactly representable small ints
    w8 = _e4m3(wv).view(torch.uint8)
    out = ops.gemm_fp8(a8.cuda(), torch.ones(K).cuda(), w8.cuda(), torch.ones(N).cuda())
    assert torch.equal(out.float().cpu(), wv.t().contiguous())


@pytest.mark.parametrize("K", [3072, 8192])
def test_gemv_fp8_step_folds_are_the_separate_launches(ops, K):
    """p3v_gemv_fp8_step (round 6, config 5): the replayed step's two ends inside the e4m3 projections next to them, against
    p3v_step_begin + p3v_gemv_fp8 and p3v_gemv_fp8 + p3v_step_end -- projection outputs, residual row, staged rotation rows, tokens
    (a tie: the first maximum; a NaN row: -1), history, both counters BIT-IDENTICAL over several steps, the ticket left at zero."""
    V, T, half, steps, B = 32064, 40, 48, 4, 1
    table = g((V, K), 170).cuda()
    cos, sin = torch.rand((B, T, half), dtype=F32).cuda(), torch.rand((B, T, half), dtype=F32).cuda()
    w1 = ops.quantize_fp8_rows((g((1024, K), 171) * 0.05).cuda())
    nw1 = (1 + 0.1 * g((K,), 172)).cuda()
    tok = torch.tensor([9001], dtype=torch.int32).cuda()
    d_past = torch.tensor([11], dtype=torch.int32).cuda()
    x_a, x_b = torch.empty((B, K), dtype=BF16).cuda(), torch.full((B, K), 7.0, dtype=BF16).cuda()
    ca, sa = torch.empty((B, 1, half), dtype=F32).cuda(), torch.empty((B, 1, half), dtype=F32).cuda()
    cb, sb = torch.zeros_like(ca), torch.zeros_like(sa)
    ops.step_begin(tok, table, x_a, cos, sin, d_past, ca, sa)
    out_a = ops.gemv_fp8(x_a, w1[0], w1[1], norm_w=nw1, norm_eps=1e-5)
    out_b = torch.full((B, 1024), float("nan"), dtype=BF16).cuda()
    assert ops.gemv_step_begin(tok, table, x_b, cos, sin, d_past, cb, sb, tuple(w1), nw1, 1e-5, out_b)
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)) and torch.equal(x_a, x_b)
    assert torch.equal(ca, cb) and torch.equal(sa, sb) and torch.equal(cb[:, 0], cos[:, 11])
    wl_bf = (g((V, K), 173) * 0.05).cuda()
    nwl = (1 + 0.1 * g((K,), 174)).cuda()
    amax_ws = torch.zeros((ops.L.GEMV_STEP_WS_BYTES // 4,), dtype=F32).cuda()      # (candidate records + arrival counters: zero once)
    hist_a, hist_b = torch.zeros((B, steps), dtype=torch.int32).cuda(), torch.zeros((B, steps), dtype=torch.int32).cuda()
    st_a, st_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    pa, pb = d_past.clone(), d_past.clone()
    tk_a, tk_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    nx_a, nx_b, to_a, to_b = (torch.zeros(B, dtype=torch.int32).cuda() for _ in range(4))
    for s in range(steps + 1):
        x = g((B, K), 190 + s).cuda()
        if s == 2:
            wl_bf[20000] = wl_bf[300]                             # a tie between two vocabulary rows: the first one wins
        if s == 3:
            x[0, 5] = float("nan")
        wl = ops.quantize_fp8_rows(wl_bf)
        lg_a = ops.gemv_fp8(x, wl[0], wl[1], norm_w=nwl, norm_eps=1e-5)
        ops.step_end(lg_a, nx_a, to_a, hist_a, st_a, pa, tk_a)
        lg_b = torch.empty((B, V), dtype=BF16).cuda()
        assert ops.gemv_step_end(x, tuple(wl), nwl, 1e-5, lg_b, nx_b, to_b, hist_b, st_b, pb, tk_b, amax_ws)
        assert torch.equal(lg_a.view(torch.int16), lg_b.view(torch.int16))
        assert torch.equal(nx_a, nx_b) and torch.equal(to_a, to_b), (s, nx_a.tolist(), nx_b.tolist())
        assert st_b.item() == s + 1 and pb.item() == 12 + s and tk_b.item() == 0
        if s == 3:
            assert nx_b[0].item() == -1
    assert torch.equal(hist_a, hist_b)
    assert not ops.gemv_step_end(g((2, K), 1).cuda(), tuple(wl), nwl, 1e-5, torch.empty((2, V), dtype=BF16).cuda(), nx_b, to_b, hist_b, st_b, pb, tk_b, amax_ws)


def test_gemv_q4_step_folds_are_the_separate_launches(ops, K=3072):
    """p3v_gemv_q4_step (round 6): the replayed step's two ends inside the MLX 4-bit projections next to them, against p3v_step_begin +
    p3v_gemv_q4 and p3v_gemv_q4 + p3v_step_end: outputs, residual row, rotation rows, tokens, history and counters BIT-IDENTICAL."""
    from phi_3_vision_mlx_amd.weights import mlx_quantize, q4_repack
    V, T, half, steps, B = 32064, 40, 48, 3, 1
    table = g((V, K), 270).cuda()
    cos, sin = torch.rand((B, T, half), dtype=F32).cuda(), torch.rand((B, T, half), dtype=F32).cuda()
    w1 = tuple(t.cuda() for t in q4_repack(*mlx_quantize(g((1024, K), 271, 0.03))))
    nw1 = (1 + 0.1 * g((K,), 272)).cuda()
    tok = torch.tensor([1234], dtype=torch.int32).cuda()
    d_past = torch.tensor([11], dtype=torch.int32).cuda()
    x_a, x_b = torch.empty((B, K), dtype=BF16).cuda(), torch.full((B, K), 7.0, dtype=BF16).cuda()
    ca, sa = torch.empty((B, 1, half), dtype=F32).cuda(), torch.empty((B, 1, half), dtype=F32).cuda()
    cb, sb = torch.zeros_like(ca), torch.zeros_like(sa)
    ops.step_begin(tok, table, x_a, cos, sin, d_past, ca, sa)
    out_a = ops.gemv_q4(x_a, w1[0], w1[1], norm_w=nw1, norm_eps=1e-5)
    out_b = torch.full((B, 1024), float("nan"), dtype=BF16).cuda()
    assert ops.gemv_step_begin(tok, table, x_b, cos, sin, d_past, cb, sb, w1, nw1, 1e-5, out_b)
    assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)) and torch.equal(x_a, x_b)
    assert torch.equal(ca, cb) and torch.equal(sa, sb)
    wl = tuple(t.cuda() for t in q4_repack(*mlx_quantize(g((V, K), 273, 0.03))))
    nwl = (1 + 0.1 * g((K,), 274)).cuda()
    amax_ws = torch.zeros((ops.L.GEMV_STEP_WS_BYTES // 4,), dtype=F32).cuda()      # (candidate records + arrival counters: zero once)
    hist_a, hist_b = torch.zeros((B, steps), dtype=torch.int32).cuda(), torch.zeros((B, steps), dtype=torch.int32).cuda()
    st_a, st_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    pa, pb = d_past.clone(), d_past.clone()
    tk_a, tk_b = torch.zeros(1, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    nx_a, nx_b, to_a, to_b = (torch.zeros(B, dtype=torch.int32).cuda() for _ in range(4))
    for s in range(steps):
        x = g((B, K), 290 + s).cuda()
        lg_a = ops.gemv_q4(x, wl[0], wl[1], norm_w=nwl, norm_eps=1e-5)
        ops.step_end(lg_a, nx_a, to_a, hist_a, st_a, pa, tk_a)
        lg_b = torch.empty((B, V), dtype=BF16).cuda()
        assert ops.gemv_step_end(x, wl, nwl, 1e-5, lg_b, nx_b, to_b, hist_b, st_b, pb, tk_b, amax_ws)
        assert torch.equal(lg_a.view(torch.int16), lg_b.view(torch.int16))
        assert torch.equal(nx_a, nx_b) and torch.equal(to_a, to_b) and st_b.item() == s + 1 and pb.item() == 12 + s and tk_b.item() == 0
    assert torch.equal(hist_a, hist_b)
