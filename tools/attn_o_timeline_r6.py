"""Round 6 timeline of the fused decode attention + o_proj launch (k_attn_decode128_o) under its role placements: the (split, head)
grid of rounds 4-5 (attn_fo_map = 0) and the virtual-CU placements (attn_fo_map = 1 / 2, fo_map in p3v_attention.hip).  Library built with
-DP3V_ATTN_TIMING (tools/build_variant.sh timing p3v_attention.hip -DP3V_ATTN_TIMING; P3V_LIB=build/libp3v_timing.so).
    python tools/attn_o_timeline_r6.py [past]"""
import os, sys, importlib, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("phi-3-vision-mlx_amd.ops")
Lm = importlib.import_module("phi-3-vision-mlx_amd._lib")
past = int(sys.argv[1]) if len(sys.argv) > 1 else 2540
nh, hd, NL, H = 32, 96, 32, 3072
Tp = (past + 128 + 127) // 128 * 128
if (Tp // 128) % 2 == 0:
    Tp += 128
n_split = Tp // 128
dev = "cuda"
torch.manual_seed(0)
kc = [torch.randn(1, nh, Tp, hd, device=dev).bfloat16() for _ in range(NL)]
vc = [torch.randn(1, nh, hd, Tp, device=dev).bfloat16() for _ in range(NL)]
wo = [(torch.randn(H, H, device=dev) * 0.02).bfloat16() for _ in range(NL)]
qkv = torch.randn(1, 1, 3 * nh * hd, device=dev).bfloat16()
cos = torch.rand(1, 1, hd // 2, device=dev); sin = torch.rand(1, 1, hd // 2, device=dev)
ws = ops.attention_ws(1, 1, nh, hd, n_split, dev)
d_past = torch.full((1,), past, device=dev, dtype=torch.int32)


def fo_map(L, n_split, nh, n_units, two_on_last=False):
    G = n_split * nh; base, r = G >> 8, G & 255
    v, slot = L & 255, L >> 8
    ns = base + (1 if v < r else 0)
    m0 = (base - 1) * 256 + 224
    if m0 <= L < m0 + 32:
        return n_split - 1, L - m0, -1
    u = L - min(max(L - m0, 0), 32)
    by, bx = divmod(u, n_split - 1)
    unit = -1
    if v < 224:
        fl = ns - 1 - slot
        idx = (v - r if v >= r else (224 - r) + v) if (base >= 2 or two_on_last) else v
        second = 224 + idx if 224 + idx < n_units else -1
        if fl == 0:
            unit = v
        elif fl == 1 and not two_on_last:
            unit = second
    return bx, by, unit


def run_mode(remap, fuse=True):
    ops.set_tuning("attn_fo_map", remap)
    o = [torch.full((1, 1, H), -1, dtype=torch.int16, device=dev).view(torch.bfloat16) for _ in range(2)]
    x = torch.zeros(1, H, device=dev, dtype=torch.bfloat16)

    def run():
        for i in range(NL):
            if fuse:
                ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], o[i & 1], 1, 1, nh, nh, hd, hd ** -0.5, past - 9, Tp, ws, n_split, d_past=d_past,
                                     merge_in_launch=True, o_proj_w=wo[i], o_proj_x=x, o_rearm=o[1 - (i & 1)])
            else:
                ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], o[0], 1, 1, nh, nh, hd, hd ** -0.5, past - 9, Tp, ws, n_split, d_past=d_past, merge_in_launch=True)
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
        per_layer = e0.elapsed_time(e1) * 1e3 / 20 / NL
    n = n_split * nh
    buf = (C.c_longlong * (n * 16))()
    assert Lm.lib().p3v_timing_read(buf, n * 16) == 0
    t = np.array(buf, dtype=np.int64).reshape(n, 16).astype(np.float64)
    us = (t - t[:, 0].min()) / 100.0
    if remap and fuse:
        roles = [fo_map(L, n_split, nh, H // 8, remap == 2) for L in range(n)]
    else:                                                       # 2-D grid: index = bx + n_split * by
        roles = [(L % n_split, L // n_split, ((L // n_split) * (n_split - 1) + L % n_split) if (L % n_split) < n_split - 1 and ((L // n_split) * (n_split - 1) + L % n_split) < H // 8 else -1)
                 for L in range(n)]
    bx = np.array([r[0] for r in roles]); unit = np.array([r[2] for r in roles])
    merger = bx == n_split - 1
    nlive = (past + 1 + 127) // 128
    work = (~merger) & (bx < nlive)
    proj = unit >= 0

    def row(name, a):
        print("  %-34s min %5.2f  mean %5.2f  p90 %5.2f  max %5.2f" % (name, a.min(), a.mean(), np.percentile(a, 90), a.max()))
    print("== attn_fo_map = %d, fused = %s: %.2f us per layer (32-layer graph, events; the timing build runs ~1 us slower than the shipped one)" % (remap, fuse, per_layer))
    row("entry (all)", us[:, 0]); row("tile DMA issued (7)", us[work, 7]); row("Q fetched / tile phase (1)", us[work, 1])
    row("P.V done (2)", us[work, 2]); row("partial stored (3)", us[work, 3])
    row("merger: all partials seen (4)", us[merger, 4]); row("merger: row stored (5)", us[merger, 5])
    if fuse:
        row("projecting: canary seen (12)", us[proj, 12]); row("projecting: dots done (13)", us[proj, 13]); row("projecting: rows stored (14)", us[proj, 14])
        print("  last partial %.2f -> last merged row %.2f -> last canary %.2f -> last o_proj row %.2f" % (
            us[work, 3].max(), us[merger, 5].max(), us[proj, 12].max(), us[proj, 14].max()))
        pw = work & proj
        print("  P.V done: projecting workgroups mean %.2f max %.2f ; others mean %.2f max %.2f" % (
            us[pw, 2].mean(), us[pw, 2].max(), us[work & ~proj, 2].mean(), us[work & ~proj, 2].max()))
    cu = t[:, 15].astype(np.int64)
    same = sum(int(cu[L] == cu[L + 256]) for L in range(n - 256))
    print("  physical CU: %d distinct; cu[L] == cu[L + 256] for %d of %d; workgroups per CU min %d max %d" % (
        len(set(cu.tolist())), same, n - 256, np.bincount(np.unique(cu, return_inverse=True)[1]).min(), np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
    if fuse:
        mcus = set(cu[merger].tolist())
        on_m = np.array([c in mcus for c in cu.tolist()])
        print("  workgroups on a merger's CU: %d, of them projecting: %d" % (int(on_m.sum()), int((on_m & proj).sum())))
        worst = int(np.argmax(np.where(merger, us[:, 5], -1)))
        mates = [L for L in range(n) if cu[L] == cu[worst]]
        print("  slowest merger L=%d head %d: partials seen %.2f row stored %.2f; CU mates %s (projecting: %s)" % (
            worst, roles[worst][1], us[worst, 4], us[worst, 5], mates, [bool(proj[L]) for L in mates]))
        ml = [L for L in range(n) if merger[L]]
        print("  mergers (head: partials seen -> row stored | tries): " + " ".join("%d:%.2f->%.2f|%d" % (roles[L][1], us[L, 4], us[L, 5], int(t[L, 6])) for L in ml))
        gap = us[:, 3] - us[:, 2]
        print("  PV done -> partial stored: mean %.2f p90 %.2f max %.2f ; of projecting workgroups' CU mates: mean %.2f max %.2f" % (
            gap[work].mean(), np.percentile(gap[work], 90), gap[work].max(),
            gap[work & ~proj & np.array([any(proj[M] for M in range(L % 256, n, 256)) for L in range(n)])].mean(),
            gap[work & ~proj & np.array([any(proj[M] for M in range(L % 256, n, 256)) for L in range(n)])].max()))
        late = int(np.argmax(np.where(work, us[:, 3], -1)))
        print("  latest partial L=%d (split %d head %d unit %d): DMA issued %.2f tile %.2f PV %.2f stored %.2f; CU mates %s" % (
            late, roles[late][0], roles[late][1], roles[late][2], us[late, 7], us[late, 1], us[late, 2], us[late, 3], [L for L in range(n) if cu[L] == cu[late]]))
    xcd = np.arange(n) % 8
    print("  by XCD: entry " + " ".join("%.2f" % us[xcd == k, 0].mean() for k in range(8)) + " | partial stored (max) " +
          " ".join("%.2f" % us[work & (xcd == k), 3].max() for k in range(8)))
    return per_layer


modes = [(2, True), (1, True), (0, True), (0, False), (2, True)]
if len(sys.argv) > 2:
    modes = [tuple(int(v) for v in m.split(",")) for m in sys.argv[2:]]
for mode in modes:
    run_mode(mode[0], bool(mode[1]))
