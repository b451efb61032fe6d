"""timeline of one k_gemv_mfma8 launch (library built with -DP3V_ATTN_TIMING)"""
import os, sys, importlib, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("phi-3-vision-mlx_amd.ops")
Lm = importlib.import_module("phi-3-vision-mlx_amd._lib")
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
N, K, epi, norm = {"qkv": (9216, 3072, ops.EPI_NONE, True), "o": (3072, 3072, ops.EPI_RESID_BF16, False),
                   "gu": (8192, 3072, ops.EPI_SILU_MUL, True), "down": (3072, 8192, ops.EPI_RESID_BF16, False),
                   "qkv_nonorm": (9216, 3072, ops.EPI_NONE, False), "small_norm": (3072, 3072, ops.EPI_NONE, True),
                   "n6144_norm": (6144, 3072, ops.EPI_NONE, True)}[which]
M = 8
rows = 2 * N if epi == ops.EPI_SILU_MUL else N
NL = 16
x = torch.randn(M, K, device="cuda").bfloat16()
Ws = [torch.randn(rows, K, device="cuda").bfloat16() * 0.02 for _ in range(NL)]
res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
nw = torch.ones(K, device="cuda").bfloat16()
kw = {}
if epi == ops.EPI_RESID_BF16: kw.update(resid=res, out=res)
if norm: kw.update(norm_w=nw, norm_eps=1e-5)
def run():
    for i in range(NL): ops.gemv(x, Ws[i], epi, **kw)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): run()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(50): g.replay()
    b.record(); torch.cuda.synchronize()
print(which, "M=8 N", N, "K", K, "us/launch %.2f" % (a.elapsed_time(b) / 50 / NL * 1e3), "MB", rows * K * 2 / 1e6)
lib = Lm.lib()
if hasattr(lib, "p3v_gemv_timing_read"):
    import os
    n_sets = -(-N // (8 if epi == ops.EPI_SILU_MUL else 16))
    per = -(-n_sets // int(os.environ.get("P3V_GEMV8_WGS", 256)))
    nwg = -(-n_sets // per)
    print("sets", n_sets, "workgroups", nwg, "sets per workgroup", per)
    buf = (C.c_longlong * (nwg * 8))()
    lib.p3v_gemv_timing_read.argtypes = [C.c_void_p, C.c_int]
    assert lib.p3v_gemv_timing_read(buf, nwg * 8) == 0
    t = np.array(buf, dtype=np.int64).reshape(nwg, 8).astype(np.float64)
    us = (t - t[:, 0].min()) / 100.0
    for k, nm in enumerate(["entry", "x loaded+parked, 2 stages issued", "3rd stage issued", "norm done (x in LDS)", "last set consumed", "-", "last set exchanged"]):
        print("%-34s min %.2f mean %.2f p90 %.2f max %.2f" % (nm, us[:, k].min(), us[:, k].mean(), np.percentile(us[:, k], 90), us[:, k].max()))
