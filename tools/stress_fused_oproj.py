"""Repeat the fused attention + o_proj parity scenario of tests/test_kernels_gpu.py many times; on a mismatch, find out WHICH launch
deviated (the two-launch path is run three times, the fused one three times) and where.  python tools/stress_fused_oproj.py [rounds]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from phi_3_vision_mlx_amd import ops
from test_kernels_gpu import g, BF16

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cases = [(2531, 2688, True), (300, 1664, False), (2559, 2688, True), (1000, 1792, True)]
B, L, nh, hd, H = 1, 1, 32, 96, 3072
stats = {}
for r in range(rounds):
    for past, cap, dev_past in cases:
        T, n_split = cap, cap // 128
        qkv = g((1, 3 * nh * hd), 145 + r).cuda()
        kc0, vc0 = g((B, nh, T, hd), 146).cuda(), g((B, nh, hd, T), 147).cuda()
        wo = (g((H, nh * hd), 148) * 0.05).cuda()
        x0 = g((1, H), 149).cuda()
        cos, sin = torch.rand((B, 1, hd // 2), device="cuda"), torch.rand((B, 1, hd // 2), device="cuda")
        d_past = torch.tensor([past], dtype=torch.int32).cuda()
        ws = ops.attention_ws(B, L, nh, hd, n_split, "cuda")
        kw = dict(d_past=d_past if dev_past else None, merge_in_launch=True)
        hp = past - 40 if dev_past else past

        def two():
            k1, v1, o1, x1 = kc0.clone(), vc0.clone(), torch.empty((1, 1, H), dtype=BF16, device="cuda"), x0.clone()
            ops.attention_decode(qkv, cos, sin, 1, k1, v1, o1, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw)
            ops.gemv(o1.view(1, H), wo, ops.EPI_RESID_BF16, resid=x1, out=x1)
            return o1.view(torch.int16).clone(), x1.view(torch.int16).clone()

        def fused():
            k2, v2, x2 = kc0.clone(), vc0.clone(), x0.clone()
            o2 = torch.full((1, 1, H), -1, dtype=torch.int16, device="cuda").view(BF16)
            other = torch.zeros((1, 1, H), dtype=BF16, device="cuda")
            ops.attention_decode(qkv, cos, sin, 1, k2, v2, o2, B, L, nh, nh, hd, hd ** -0.5, hp, T, ws, n_split, **kw,
                                 o_proj_w=wo, o_proj_x=x2, o_rearm=other)
            return o2.view(torch.int16).clone(), x2.view(torch.int16).clone()

        runs = [("two", two()), ("fused", fused()), ("two", two()), ("fused", fused()), ("two", two()), ("fused", fused())]
        ref_o = torch.stack([o for _, (o, _) in runs]).mode(0).values          # majority vote per word
        ref_x = torch.stack([x for _, (_, x) in runs]).mode(0).values
        for idx, (name, (o, x)) in enumerate(runs):
            bad_o, bad_x = (o != ref_o).flatten(), (x != ref_x).flatten()
            if bad_o.any() or bad_x.any():
                key = (name, idx)
                stats[key] = stats.get(key, 0) + 1
                w = bad_o.nonzero().flatten().tolist()
                print(f"round {r} case {(past, cap, dev_past)} run {idx} ({name}): {len(w)} attention words off, heads {sorted(set(i // hd for i in w))},"
                      f" residual words off {int(bad_x.sum())}; first values {[(int(o.flatten()[i]), int(ref_o.flatten()[i])) for i in w[:4]]}", flush=True)
    if r % 50 == 49:
        print(f"round {r + 1}: deviating runs by (kind, position) {stats}", flush=True)
print("total deviating runs by (kind, position in the sequence two/fused/two/fused/two/fused):", stats, "over", rounds * len(cases), "sequences")
