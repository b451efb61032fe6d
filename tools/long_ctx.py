"""BASELINE config 3 smoke: 32k-token prefill + decode on the full-size blind model (Su/LongRoPE long factors)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
model, _ = load_synthetic(blind_model=True, device="cuda:0", use_quantized_cache=bool(os.environ.get("P3V_QCACHE")), quantized_fp8=bool(os.environ.get("P3V_FP8")))
ids = np.random.default_rng(0).integers(3, 32000, (1, S))
lg = cache = None
for rep in range(3):
    lg = cache = None                                          # release the previous cache first: the timed call must reuse its blocks
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lg, cache = model(input_ids=ids, max_tokens=int(sys.argv[2]) if len(sys.argv) > 2 else 136)
    tok = ops.argmax(lg[:, -1].contiguous())[:, None]; tok.tolist()
    pre = time.perf_counter() - t0
print(f"S={S}: prefill {pre*1e3:.1f} ms  ({(2*S*3.722e9 + 2*32*S*S*3072)/pre/1e12:.0f} TFLOP/s algorithmic)  mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
t = tok
for _ in range(8): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(64): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 64
kv = 2 * 32 * 32 * 96 * 2 * (S + 40)
print(f"decode {dt*1e3:.3f} ms/step = {1/dt:.1f} tok/s; bytes/token {(7.445e9+kv)/1e9:.2f} GB -> {(7.445e9+kv)/dt/1e12:.2f} TB/s")
