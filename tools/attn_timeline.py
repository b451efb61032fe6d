"""timeline of one k_attn_decode128 launch (library built with -DP3V_ATTN_TIMING): per-workgroup 100 MHz timestamps"""
import os, sys, importlib, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("phi-3-vision-mlx_amd.ops")
Lm = importlib.import_module("phi-3-vision-mlx_amd._lib")
past = int(sys.argv[1]) if len(sys.argv) > 1 else 2540
nh, hd, NL = 32, 96, 32
Tp = (past + 24 + 127) // 128 * 128
n_split = Tp // 128
dev = "cuda"
torch.manual_seed(0)
kc = [torch.randn(1, nh, Tp, hd, device=dev).bfloat16() for _ in range(NL)]
vc = [torch.randn(1, nh, hd, Tp, device=dev).bfloat16() for _ in range(NL)]
qkv = torch.randn(1, 1, 3 * nh * hd, device=dev).bfloat16()
cos = torch.rand(1, 1, hd // 2, device=dev); sin = torch.rand(1, 1, hd // 2, device=dev)
out = torch.empty(1, 1, nh * hd, device=dev, dtype=torch.bfloat16)
ws = ops.attention_ws(1, 1, nh, hd, n_split, dev)
d_past = torch.full((1,), past, device=dev, dtype=torch.int32)
def run():
    for i in range(NL):
        ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], out, 1, 1, nh, nh, hd, hd ** -0.5, past, Tp, ws, n_split, d_past=d_past, merge_in_launch=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): run()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
n = n_split * nh
buf = (C.c_longlong * (n * 16))()
lib = Lm.lib()
assert lib.p3v_timing_read(buf, n * 16) == 0
t = np.array(buf, dtype=np.int64).reshape(nh, n_split, 16).astype(np.float64)
t0 = t[:, :, 0].min()
us = (t - t0) / 100.0
nm = ["entry", "tile landed", "PV done", "partial stored", "poll done", "out stored"]
merger = n_split - 1
work = us[:, :merger - 1] if past < (n_split - 1) * 128 else us[:, :merger]   # splits with keys
print("splits", n_split, "past", past)
for k in range(4):
    x = us[:, :, k] if k in (0, 3) else work[:, :, k]
    print("%-15s min %.2f  mean %.2f  p90 %.2f  max %.2f" % (nm[k], x.min(), x.mean(), np.percentile(x, 90), x.max()))
for k, name in ((7, "loads issued"), (8, "Q in LDS"), (9, "mark cost"), (10, "softmax done"), (11, "PV issued+used")):
    x = work[:, :, k]
    print("%-15s min %.2f  mean %.2f  max %.2f   (delta to previous phase mean)" % (name, x.min(), x.mean(), x.max()))
d = work
print("phase means: entry->loads issued %.2f | ->Q in LDS %.2f | ->landed+barrier %.2f | mark %.2f | ->softmax %.2f | ->PV %.2f | ->barrier %.2f | ->stored %.2f" % (
    (d[:, :, 7] - d[:, :, 0]).mean(), (d[:, :, 8] - d[:, :, 7]).mean(), (d[:, :, 1] - d[:, :, 8]).mean(), (d[:, :, 9] - d[:, :, 1]).mean(),
    (d[:, :, 10] - d[:, :, 9]).mean(), (d[:, :, 11] - d[:, :, 10]).mean(), (d[:, :, 2] - d[:, :, 11]).mean(), (d[:, :, 3] - d[:, :, 2]).mean()))
for k in (4, 5):
    x = us[:, merger, k]
    print("%-15s min %.2f  mean %.2f  max %.2f   (merger workgroups)" % (nm[k], x.min(), x.mean(), x.max()))
print("poll passes: mean %.2f max %d" % (t[:, merger, 6].mean(), t[:, merger, 6].max()))
last_partial = us[:, :merger, 3].max(axis=1)          # per head: when its slowest split stored
print("per head: slowest split's store -> merger poll done: mean %.2f us; merger poll done -> out stored: mean %.2f us"
      % ((us[:, merger, 4] - last_partial).mean(), (us[:, merger, 5] - us[:, merger, 4]).mean()))
print("last out stored at %.2f us after the first workgroup's entry" % us[:, merger, 5].max())
