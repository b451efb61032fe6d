"""Cycle-counter breakdown of one k_attn_decode block (debug build: -DP3V_ATTN_DEBUG -> gpurun_out/libp3v_dbg.so (built by hand, never committed))."""
import sys, os, ctypes
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "libp3v_dbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops, _lib
from phi_3_vision_mlx_amd.api import load_synthetic
model, _ = load_synthetic(blind_model=True, device="cuda:0")
ids = np.random.default_rng(0).integers(3, 32000, (1, 2531))
lg, cache = model(input_ids=ids, max_tokens=40)
t = ops.argmax(lg[:, -1].contiguous())[:, None]
for _ in range(10): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L = _lib.lib(); L.p3v_debug_read.argtypes = [ctypes.c_void_p]; L.p3v_debug_read(buf)
v = list(buf)
order = [(0, "entry"), (9, "tile DMA issued"), (2, "q rotated"), (1, "past arrived"), (3, "tile landed"), (4, "S^T MFMAs done"), (5, "softmax done"), (6, "PV done"), (7, "partials stored")]
for (i, n), (j, _) in zip(order[1:], order[:-1]): print(f"{n:20s} +{v[i]-v[j]:6d} cycles  (cum {v[i]-v[0]})")
print(f"launch timeline (100 MHz clock): last workgroup enters +{(v[12]-v[10])/100:.2f} us after the first, its partials stored +{(v[13]-v[10])/100:.2f} us, merge done +{(v[11]-v[10])/100:.2f} us")
