"""Phase timeline of ONE workgroup of k_attn_prefill_pp (debug build: csrc/build.sh -DP3V_PP_DEBUG=<block> -o ../../build/libp3v_ppdbg.so).
Per wave: the SIMD it ran on (HW_ID) and, per step, cycles spent in its phase and waiting at the step's barrier."""
import ctypes, os, sys
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{os.environ.get('PP_VARIANT', 'ppdbg')}.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops, _lib
L, nh, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 32, 96
q = (torch.randn(1, nh, L, hd, device="cuda") * (hd ** -0.5 * ops.Q_PRESCALE)).bfloat16()
k = torch.randn(1, nh, L, hd, device="cuda").bfloat16()
v = torch.randn(1, nh, hd, L, device="cuda").bfloat16()
out = torch.empty(1, L, nh * hd, device="cuda", dtype=torch.bfloat16)
ops.set_tuning("attn_pp", 1)
for _ in range(2):
    ops.attention(q, out, 1, L, nh, nh, hd, hd ** -0.5, True, k_past=k, v_past=v, past_t=L, new_is_cache=True, q_prescaled=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 128))()
lib = _lib.lib()
lib.p3v_ppdbg_read.argtypes = [ctypes.c_void_p]
lib.p3v_ppdbg_read(buf)
t = [list(buf[w * 128:(w + 1) * 128]) for w in range(8)]
for w in range(8):
    hw = t[w][0]
    print(f"wave {w}: HW_ID wave {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}")
t0 = min(t[w][1] for w in range(8))
print("step: per wave  phase cycles / barrier-wait cycles   (group A = waves 0-3: even steps = matrix phase; group B one step behind)")
print("variant", os.environ.get("PP_VARIANT", "ppdbg"))
for step in range(8, 14):
    row = []
    for w in (0, 1, 4, 5):
        a, b, c = t[w][1 + 2 * step], t[w][2 + 2 * step], t[w][3 + 2 * step]
        row.append(f"w{w} {b - a:5d}/{c - b:5d}")
    print(f"  step {step:2d} (+{t[0][1 + 2 * step] - t0:7d})  " + "   ".join(row))

buf2 = (ctypes.c_ulonglong * (8 * 64 * 4))()
if hasattr(lib, "p3v_ppdbg2_read"):
    lib.p3v_ppdbg2_read.argtypes = [ctypes.c_void_p]
    lib.p3v_ppdbg2_read(buf2)
    print("inside a phase: cycles from the step's start to the four stamps ; phase end.  Matrix phase (A: even steps, B: odd): before the fragment reads | reads issued | PV issued | S^T issued.  VALU phase: DMA issued | maxima reduced | exponentials of both halves issued (packing of the first too) | P packed")
    for step in range(8, 14):
        row = []
        for w in (0, 4):
            st = [buf2[(w * 64 + step) * 4 + k] for k in range(4)]
            a, b = t[w][1 + 2 * step], t[w][2 + 2 * step]
            if st[0] >= a and st[3] <= b + 64: row.append(f"w{w} " + " ".join(f"{x - a:5d}" for x in st) + f" ;{b - a:5d}")
        print(f"  step {step:2d}  " + "   ".join(row))
