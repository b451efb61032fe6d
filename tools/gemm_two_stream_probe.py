"""Would the leading-rows (256 x 256 tiles) and remainder (128 x 128 tiles) launches of a prompt-sized projection gain from
running on two streams?  Same two sub-problems, one stream against two (fork / join with events), alternated."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
for (M, N, K, big) in ((2531, 9216, 3072, 1792), (2531, 16384, 3072, 2048)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    o1 = torch.empty(big, N, device="cuda", dtype=torch.bfloat16); o2 = torch.empty(M - big, N, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    def seq():
        ops.gemm(x[:big], w, ops.EPI_NONE, out=o1); ops.gemm(x[big:], w, ops.EPI_NONE, out=o2)
    def par():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm(x[big:], w, ops.EPI_NONE, out=o2)
            done = torch.cuda.Event(); done.record()
        ops.gemm(x[:big], w, ops.EPI_NONE, out=o1)
        torch.cuda.current_stream().wait_event(done)
    def whole():
        ops.gemm(x, w, ops.EPI_NONE)
    t = {"whole": [], "seq": [], "par": []}
    for rep in range(7):
        for name, f in (("whole", whole), ("seq", seq), ("par", par)):
            f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize(); t[name].append(a.elapsed_time(b) / 20 * 1e3)
    print(f"M={M} N={N} K={K} (leading {big} rows): " + "   ".join(f"{k_} {statistics.median(v):7.1f} us" for k_, v in t.items()), flush=True)
