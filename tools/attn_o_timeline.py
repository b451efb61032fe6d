"""timeline of one k_attn_decode128_o launch (library built with -DP3V_ATTN_TIMING): when the workers see the canary, have the
attention vector and have stored their o_proj rows.  python tools/attn_o_timeline.py [past]"""
import os, sys, importlib, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("phi-3-vision-mlx_amd.ops")
Lm = importlib.import_module("phi-3-vision-mlx_amd._lib")
past = int(sys.argv[1]) if len(sys.argv) > 1 else 2540
fuse = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
nh, hd, NL, H = 32, 96, 32, 3072
Tp = (past + 24 + 127) // 128 * 128
n_split = Tp // 128
dev = "cuda"
torch.manual_seed(0)
kc = [torch.randn(1, nh, Tp, hd, device=dev).bfloat16() for _ in range(NL)]
vc = [torch.randn(1, nh, hd, Tp, device=dev).bfloat16() for _ in range(NL)]
wo = [(torch.randn(H, H, device=dev) * 0.02).bfloat16() for _ in range(NL)]
qkv = torch.randn(1, 1, 3 * nh * hd, device=dev).bfloat16()
cos = torch.rand(1, 1, hd // 2, device=dev); sin = torch.rand(1, 1, hd // 2, device=dev)
o = [torch.full((1, 1, H), -1, dtype=torch.int16, device=dev).view(torch.bfloat16) for _ in range(2)]
x = torch.zeros(1, H, device=dev, dtype=torch.bfloat16)
ws = ops.attention_ws(1, 1, nh, hd, n_split, dev)
d_past = torch.full((1,), past, device=dev, dtype=torch.int32)
def run():
    for i in range(NL):
        if fuse:
            ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], o[i & 1], 1, 1, nh, nh, hd, hd ** -0.5, past, Tp, ws, n_split, d_past=d_past,
                                 merge_in_launch=True, o_proj_w=wo[i], o_proj_x=x, o_rearm=o[1 - (i & 1)])
        else:
            ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], o[0], 1, 1, nh, nh, hd, hd ** -0.5, past, Tp, ws, n_split, d_past=d_past, merge_in_launch=True)
            ops.gemv(o[0].view(1, H), wo[i], ops.EPI_RESID_BF16, resid=x, out=x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): run()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): g.replay()
    e1.record(s); torch.cuda.synchronize()
    print("fuse", fuse, "per layer %.2f us" % (e0.elapsed_time(e1) * 1e3 / 20 / NL))
n = n_split * nh
buf = (C.c_longlong * (n * 16))()
assert Lm.lib().p3v_timing_read(buf, n * 16) == 0
t = np.array(buf, dtype=np.int64).reshape(nh, n_split, 16).astype(np.float64)
us = (t - t[:, :, 0].min()) / 100.0
merger = n_split - 1
nlive = (past + 1 + 127) // 128
work = us[:, :min(nlive, merger)]
def row(name, a): print("%-28s min %.2f  mean %.2f  p90 %.2f  max %.2f" % (name, a.min(), a.mean(), np.percentile(a, 90), a.max()))
row("entry (all)", us[:, :, 0]); row("worker: tile landed", work[:, :, 1]); row("worker: PV done", work[:, :, 2]); row("worker: partial stored", work[:, :, 3])
row("merger: poll done", us[:, merger, 4]); row("merger: out stored", us[:, merger, 5])
if fuse:
    rank = np.arange(nh)[:, None] * merger + np.arange(merger)[None, :]      # the first H / 8 non-merging workgroups project
    pw = us[:, :merger][rank < H // 8]
    row("projecting: canary seen", pw[:, 12]); row("projecting: vector in LDS", pw[:, 13]); row("projecting: rows stored", pw[:, 14])
    print("means: partial stored -> canary %.2f | canary -> vector %.2f | vector -> stored %.2f" % (
        (pw[:, 12] - pw[:, 3]).mean(), (pw[:, 13] - pw[:, 12]).mean(), (pw[:, 14] - pw[:, 13]).mean()))
    print("last merger out stored %.2f ; last projecting workgroup done %.2f" % (us[:, merger, 5].max(), pw[:, 14].max()))
# ---- round 5: where do the late workgroups lose their time?
row("worker: loads issued (7)", work[:, :, 7]); row("worker: Q staged (8)", work[:, :, 8])
d = work[:, :, 2] - work[:, :, 1]
row("worker: landed -> PV done", d)
late = work[:, :, 2] > np.percentile(work[:, :, 2], 90)
print("late (p90+) PV-done workgroups: entry %.2f  loads issued %.2f  landed %.2f  PV %.2f ; the others: %.2f %.2f %.2f %.2f" % (
    work[:, :, 0][late].mean(), work[:, :, 7][late].mean(), work[:, :, 1][late].mean(), work[:, :, 2][late].mean(),
    work[:, :, 0][~late].mean(), work[:, :, 7][~late].mean(), work[:, :, 1][~late].mean(), work[:, :, 2][~late].mean()))
print("late ones by split index (share of each split that is late):", np.round(late.mean(axis=0), 2).tolist())
print("late ones by head (share):", np.round(late.mean(axis=1), 2).tolist())
if fuse:
    proj = (rank < H // 8)[:, :work.shape[1]]
    print("PV done: projecting workgroups mean %.2f max %.2f ; non-projecting mean %.2f max %.2f" % (
        work[:, :, 2][proj].mean(), work[:, :, 2][proj].max(), work[:, :, 2][~proj].mean(), work[:, :, 2][~proj].max()))
