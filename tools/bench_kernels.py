"""Per-shape micro-benchmark of the projection kernels (run on the GPU box):
HIP-event timing over rotating operand copies (so the 256 MiB Infinity Cache cannot hold them)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops

def timeit(fn, n_rot, iters=20):
    for i in range(3): fn(i % n_rot)
    torch.cuda.synchronize()
    a, b = ops.Event(), ops.Event()
    a.record()
    for i in range(iters): fn(i % n_rot)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_ms(b) / iters

def gemm_case(name, M, N, K, epi, f32res=False):
    nrot = max(2, int(600e6 // (N * K * 2 * (2 if epi == ops.EPI_SILU_MUL else 1))) + 1)
    nrot = min(nrot, 8)
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn((2 * N if epi == ops.EPI_SILU_MUL else N), K, device="cuda").bfloat16() * 0.02 for _ in range(nrot)]
    bias = torch.randn(N, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32res else torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BIAS, ops.EPI_BIAS_QGELU, ops.EPI_BIAS_GELU, ops.EPI_BIAS_RESID_F32): kw["bias"] = bias
    if epi in (ops.EPI_BIAS_RESID_F32, ops.EPI_RESID_BF16): kw.update(resid=res, out=res)
    for persist in (0, 1):            # 256x256-tile kernel: one workgroup per tile / persistent workgroups with cross-tile prefetch
        old = ops.set_tuning("gemm_persistent", persist)
        ms = timeit(lambda i: ops.gemm(A, Ws[i], epi, **kw), nrot)
        ops.set_tuning("gemm_persistent", old)
        tf = 2 * M * N * K * (2 if epi == ops.EPI_SILU_MUL else 1) / ms / 1e9
        print(f"gemm {name:28s} {'persistent' if persist else 'per-tile  '} M={M:6d} N={N:6d} K={K:5d}  {ms*1e3:9.1f} us  {tf:8.1f} TF/s", flush=True)

def gemv_case(name, M, N, K, epi, norm=False):
    rows = 2 * N if epi == ops.EPI_SILU_MUL else N
    nrot = min(16, max(2, int(600e6 // (rows * K * 2)) + 1))
    x = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(rows, K, device="cuda").bfloat16() * 0.02 for _ in range(nrot)]
    res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    nw = torch.ones(K, device="cuda").bfloat16()
    kw = {}
    if epi == ops.EPI_RESID_BF16: kw.update(resid=res, out=res)
    if norm: kw.update(norm_w=nw, norm_eps=1e-5)
    ms = timeit(lambda i: ops.gemv(x, Ws[i], epi, **kw), nrot, iters=50)
    gb = rows * K * 2 / ms / 1e6
    print(f"gemv {name:28s} M={M:6d} N={N:6d} K={K:5d}  {ms*1e3:9.1f} us  {gb:8.1f} GB/s  ({gb/80:.1f}% of 8 TB/s)")

def attn_case(name, B, L, nh, hd, causal, prescaled=None):
    """The prompt-sized kernels, pinned and interleaved in one process (attn_pp / attn_il: dma = k_attn_prefill_dma, 128 queries per
    workgroup; pingpong = k_attn_prefill_pp; interleaved = k_attn_prefill_il, pre-scaled q only) on random data; prescaled: q
    carries scale * log2(e) (the decoder's prefill path)."""
    Tp = (L + 63) // 64 * 64
    prescaled = causal if prescaled is None else prescaled
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE if prescaled else 1.0)).bfloat16()
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16()
    v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    fl = 4 * B * nh * L * L * hd * (0.5 if causal else 1.0)
    kinds = [("dma", 0, 0), ("pingpong", 1, 0)] + ([("interleaved", 1, 1)] if prescaled else [])
    t = {n: [] for n, _, _ in kinds}
    for rep in range(3):                                    # the chip's clock follows its thermal state: alternate the kernels
        for n, pp, il in kinds:
            old, old_il = ops.set_tuning("attn_pp", pp), ops.set_tuning("attn_il", il)
            t[n].append(timeit(lambda i: ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, causal, k_past=k, v_past=v, past_t=Tp, new_is_cache=True,
                                                       q_prescaled=prescaled), 1, iters=10 if L <= 8192 else 3))
            ops.set_tuning("attn_pp", old), ops.set_tuning("attn_il", old_il)
    for n, _, _ in kinds:
        ms = sorted(t[n])[1]
        print(f"attn {name:28s} {n:11s} B={B:3d} L={L:5d} heads={nh} hd={hd}  {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "gemm"):
        S = 2531
        gemm_case("dec qkv", S, 9216, 3072, ops.EPI_NONE)
        gemm_case("dec o_proj+resid", S, 3072, 3072, ops.EPI_RESID_BF16)
        gemm_case("dec gate_up silu", S, 8192, 3072, ops.EPI_SILU_MUL)
        gemm_case("dec down+resid", S, 3072, 8192, ops.EPI_RESID_BF16)
        V = 17 * 577
        gemm_case("vit qkv+bias", V, 3072, 1024, ops.EPI_BIAS)
        gemm_case("vit out+resid_f32", V, 1024, 1024, ops.EPI_BIAS_RESID_F32, True)
        gemm_case("vit fc1 qgelu", V, 4096, 1024, ops.EPI_BIAS_QGELU)
        gemm_case("vit fc2+resid_f32", V, 1024, 4096, ops.EPI_BIAS_RESID_F32, True)
        gemm_case("proj0 gelu", 2509, 3072, 4096, ops.EPI_BIAS_GELU)
        gemm_case("square 4096", 4096, 4096, 4096, ops.EPI_NONE)
        gemm_case("square 8192", 8192, 8192, 8192, ops.EPI_NONE)
    if which in ("all", "attn"):
        attn_case("decoder prefill causal", 1, 2531, 32, 96, True)
        attn_case("decoder prefill 8k causal", 1, 8192, 32, 96, True)
        attn_case("decoder prefill 32k causal", 1, 32768, 32, 96, True)
        attn_case("decoder prefill 1k causal", 1, 1024, 32, 96, True)
        attn_case("decoder prefill 1280 causal", 1, 1280, 32, 96, True)
        attn_case("decoder prefill 8 x 512", 8, 512, 32, 96, True)
        attn_case("clip 17 crops", 17, 577, 16, 64, False)
    if which in ("all", "gemv"):
        for M in (1, 8):
            gemv_case("qkv +norm", M, 9216, 3072, ops.EPI_NONE, True)
            gemv_case("o_proj +resid", M, 3072, 3072, ops.EPI_RESID_BF16)
            gemv_case("gate_up silu +norm", M, 8192, 3072, ops.EPI_SILU_MUL, True)
            gemv_case("down +resid", M, 3072, 8192, ops.EPI_RESID_BF16)
            gemv_case("lm_head +norm", M, 32064, 3072, ops.EPI_NONE, True)
