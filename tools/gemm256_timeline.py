"""Timeline of ONE workgroup of k_gemm256's K loop (debug build: tools/build_variant.sh g256dbg p3v_gemm256.hip -DP3V_G256_DEBUG=<block>).
Per wave and K-tile, cycles from the K-tile's start to: own DMA landed (vmcnt) | barrier passed | first 12 fragment reads + 4 DMA
pieces issued | quad 0 issued | 8 more reads + 4 DMA pieces issued | quad 1 issued | quads 2, 3 issued; length of the K-tile."""
import ctypes, os, sys
os.environ["P3V_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", f"libp3v_{os.environ.get('G_VARIANT', 'g256dbg')}.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops, _lib
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
for _ in range(3): out = ops.gemm(a, w, ops.EPI_NONE)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
lib = _lib.lib()
lib.p3v_g256dbg_read.argtypes = [ctypes.c_void_p]
lib.p3v_g256dbg_read(buf)
t = lambda w_, it, k: buf[(w_ * 64 + it) * 8 + k]
print(f"M={M} N={N} K={K}: K-tile: per wave  vmcnt | barrier | reads+DMA 0 | quad 0 | reads+DMA 1 | quad 1 | quads 2,3 ; K-tile length (cycles)")
for it in range(20, 26):
    for w_ in (0, 1, 4, 5):
        a0 = t(w_, it, 0)
        print(f"  kt {it:2d} w{w_}: " + " ".join(f"{t(w_, it, k) - a0:5d}" for k in range(1, 8)) + f" ; {t(w_, it + 1, 0) - a0:5d}")
