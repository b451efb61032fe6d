"""Timeline of ONE workgroup of k_attn_prefill_il (debug build: tools/build_variant.sh ildbg p3v_attention.hip -DP3V_IL_DEBUG=<block>).
Per wave and tile iteration, cycles from the iteration's start to: end of region 1 of slot A | end of slot A | barrier passed |
DMA issued | end of region 1 of slot B | end of slot B; and the length of the iteration."""
import ctypes, os, sys
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{os.environ.get('IL_VARIANT', 'ildbg')}.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops, _lib
L, nh, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 32, 96
q = (torch.randn(1, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
k = torch.randn(1, nh, L, hd, device="cuda").bfloat16()
v = torch.randn(1, nh, hd, L, device="cuda").bfloat16()
out = torch.empty(1, L, nh * hd, device="cuda", dtype=torch.bfloat16)
ops.set_tuning("attn_il", 1)
for _ in range(2):
    ops.attention(q, out, 1, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=L, new_is_cache=True, q_prescaled=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
lib = _lib.lib()
lib.p3v_ildbg_read.argtypes = [ctypes.c_void_p]
lib.p3v_ildbg_read(buf)
t = lambda w, it, k: buf[(w * 64 + it) * 8 + k]
print("variant", os.environ.get("IL_VARIANT", "ildbg"), "L", L)
print("iteration: per wave  A.region1 | A.region2 after 4 and 8 of its 12 MFMAs | A.end | barrier | B.region1 | B.end   (cycles from the iteration's start) ; iteration length")
for it in range(8, 14):
    for w in (0, 1, 4, 5):
        a = t(w, it, 0)
        nxt = t(w, it + 1, 0)
        print(f"  it {it:2d} w{w}: " + " ".join(f"{t(w, it, k) - a:5d}" for k in (1, 4, 7, 2, 3, 5, 6)) + f" ; {nxt - a:5d}")
