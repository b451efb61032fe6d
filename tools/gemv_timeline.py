"""Per-wave entry / exit stamps of the decode GEMVs (library built with -DP3V_GEMV_TIMING): how far apart do the waves of one
launch finish?  python tools/gemv_timeline.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops, _lib
H, I, NL = 3072, 8192, 16
dev = "cuda"
torch.manual_seed(0)
gw = torch.ones(H, device=dev).bfloat16()
shapes = {"gate_up (SiLU*up, 100.7 MB)": (2 * I, H, ops.EPI_SILU_MUL), "qkv (56.6 MB)": (3 * H, H, ops.EPI_NONE),
          "down (+resid, 50.3 MB)": (H, I, ops.EPI_RESID_BF16), "o_proj (+resid, 18.9 MB)": (H, H, ops.EPI_RESID_BF16)}
lib = _lib.lib()
lib.p3v_gemv_timing_read.restype = C.c_int
for name, (N, K, epi) in shapes.items():
    ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NL)]
    x = torch.randn(1, K, device=dev).bfloat16()
    res = torch.zeros(1, N if epi != ops.EPI_SILU_MUL else N // 2, device=dev).bfloat16()
    out = torch.empty_like(res)
    def run():
        for w in ws:
            ops.gemv(x, w, epi, resid=res if epi == ops.EPI_RESID_BF16 else None, norm_w=gw if K == H and epi != ops.EPI_RESID_BF16 else None,
                     norm_eps=1e-5, out=out)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s): run()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): g.replay()
        e1.record(s); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e3 / 20 / NL
    buf = (C.c_longlong * 8192)()
    assert lib.p3v_gemv_timing_read(buf, 8192) == 0
    t = np.array(buf, dtype=np.int64).reshape(4096, 2).astype(np.float64)
    full = t.copy()
    live = (full[:, 0] > 0) & (full[:, 1] > full[:, 0] + 100)       # (wave 3 of a 3-wave workgroup takes no rows: skip it)
    t = t[live]
    t = t[t[:, 0] > t[:, 0].max() - 3000]                       # the last launch's waves
    t0 = t[:, 0].min()
    a, b = (t[:, 0] - t0) / 100, (t[:, 1] - t0) / 100
    print("%-28s %6.2f us per launch | %4d waves | entry mean %.2f max %.2f | exit min %.2f  p10 %.2f  mean %.2f  p90 %.2f  max %.2f" % (
        name, per, len(t), a.mean(), a.max(), b.min(), np.percentile(b, 10), b.mean(), np.percentile(b, 90), b.max()))
    del ws
    idx = np.nonzero(live)[0]
    idx = idx[full[idx, 0] > full[idx, 0].max() - 3000]
    nw = len(idx)
    ex = (full[idx, 1] - t0) / 100
    g = idx // 4
    print("    exit by XCD (g % 8):  " + "  ".join("%.2f" % ex[(g % 8) == k].mean() for k in range(8)))
    print("    exit by wave of the workgroup:  " + "  ".join("%.2f" % ex[idx % 4 == k].mean() for k in range(4) if (idx % 4 == k).any()))
    ng = g.max() + 1
    print("    exit by workgroup octile (g * 8 // n):  " + "  ".join("%.2f" % ex[(g * 8 // ng) == k].mean() for k in range(8)))
    print("    workgroups %d" % ng)
    order = np.argsort(-ex)[:24]
    print("    slowest waves (workgroup:wave exit): " + "  ".join("%d:%d %.1f" % (g[i], idx[i] % 4, ex[i]) for i in order))
    wgm = np.array([ex[g == k].max() for k in range(ng)])
    cu = np.arange(ng) % 256                                  # workgroup k and k + 256 share a CU if the dispatch is breadth-first
    both = np.array([wgm[cu == c].max() for c in range(min(256, ng))])
    print("    per CU (last exit of its workgroups): min %.2f  mean %.2f  p90 %.2f  max %.2f; by XCD (cu %% 8): " % (both.min(), both.mean(), np.percentile(both, 90), both.max())
          + "  ".join("%.2f" % both[np.arange(len(both)) % 8 == k].mean() for k in range(8)))
