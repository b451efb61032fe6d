import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phi_3_vision_mlx_amd.api import load_synthetic
model, _ = load_synthetic(blind_model=True, device="cuda:0")
for S in (17, 32, 64, 128, 256, 512, 1024):
    ids = np.random.default_rng(0).integers(3, 32000, (1, S))
    for _ in range(2): model.greedy_prefill(8, input_ids=ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): model.greedy_prefill(8, input_ids=ids)
    torch.cuda.synchronize(); print(f"S={S:5d}: prefill {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
