"""decode attention in a graph of 32 launches over rotating caches: fused merge vs partial-only (+ combine kernel)"""
import os, sys, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("phi-3-vision-mlx_amd.ops")
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
past = int(sys.argv[2]) if len(sys.argv) > 2 else 2540
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
        ops.attention_decode(qkv, cos, sin, 1, kc[i], vc[i], out, 1, 1, nh, nh, hd, hd ** -0.5, past, Tp, ws, n_split, d_past=d_past, merge_in_launch=(mode == "fused"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s): run()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(50): g.replay()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 / NL * 1e3)
import os
ref = torch.empty_like(out)
print(mode, "xcd", os.environ.get("P3V_ATTN_XCD"), "past", past, "Tp", Tp, "n_split", n_split, "us/launch min %.2f med %.2f" % (min(ts), sorted(ts)[3]),
      "nan" if torch.isnan(out.float()).any() else "ok", float(out.float().abs().sum()))
