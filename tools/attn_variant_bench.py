"""Kernel time of k_attn_prefill_pp builds (tools/build_variant.sh <name> p3v_attention.hip -D...) -- several builds INTERLEAVED in
one process (the chip's clock follows its thermal state: back-to-back processes are not comparable).
  python tools/attn_variant_bench.py v1,v2[:il] [shape ...]     shape = B:L:heads:hd:causal; ":il" = k_attn_prefill_il of that build"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import _lib, ops
names = sys.argv[1].split(",")
shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(1, 8192, 32, 96, 1)]
libs = {}
for n in names:
    l = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{n.split(':')[0]}.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    l.p3v_set_tuning(b"attn_pp", int(not n.endswith(":dma")))      # "name:dma": the 128-query kernel of that build
    l.p3v_set_tuning(b"attn_il", int(n.endswith(":il")))          # "name:il": the interleaved kernel of that build
    libs[n] = l
for B, L, nh, hd, causal in shapes:
    Tp = (L + 63) // 64 * 64
    pre = bool(causal)
    q = (torch.randn(B, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE if pre else 1.0)).bfloat16()
    k = torch.randn(B, nh, Tp, hd, device="cuda").bfloat16()
    v = torch.randn(B, nh, hd, Tp, device="cuda").bfloat16()
    out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
    t = {n: [] for n in names}
    for rep in range(5):
        for n in names:
            _lib._lib = libs[n]
            f = lambda: ops.attention(q, out, B, L, nh, nh, hd, hd ** -0.5, bool(causal), k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
            f(); torch.cuda.synchronize()
            a, b = ops.Event(), ops.Event()
            a.record()
            for _ in range(5 if L > 16384 else 10): f()
            b.record(); torch.cuda.synchronize()
            t[n].append(a.elapsed_ms(b) / (5 if L > 16384 else 10))
    fl = 4 * B * nh * L * L * hd * (0.5 if causal else 1.0)
    print(f"B={B} L={L} heads={nh} hd={hd} causal={causal}: " + "   ".join(f"{n} {statistics.median(t[n])*1e3:8.1f} us ({fl/statistics.median(t[n])/1e9:6.0f} TF/s)" for n in names), flush=True)
