"""BASELINE config 4, one GPU's share: B image+text prompts (default 8) through the model call generate() uses --
B 336x336 images (17 crops each) preprocessed on the device, one prefill over B x 2531 tokens, graph-replayed greedy decode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, proc = load_synthetic(blind_model=False, seed=0, device="cuda:0")
rng = np.random.default_rng(0)
imgs = [Image.fromarray(rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)) for _ in range(B)]
proc.img_processor.device_call(imgs, "cuda:0"); torch.cuda.synchronize()
t0 = time.perf_counter()
im = proc.img_processor.device_call(imgs, "cuda:0"); torch.cuda.synchronize()
print(f"B={B} device preprocessing of {B} images: {(time.perf_counter() - t0) * 1e3:.1f} ms")
n_img = im["num_img_tokens"][0]
rows = []
for b in range(B):
    t = rng.integers(3, 32000, 20)
    rows.append(np.concatenate([[1], t[:8], -np.ones(n_img, dtype=np.int64), [1], t[8:]]))
ids = np.stack(rows).astype(np.int64)
inp = {"input_ids": ids, "pixel_values": im["pixel_values"], "image_sizes": np.asarray(im["image_sizes"]), "positions": np.argwhere(ids < 0)}
def prefill():
    lg, cache = model(**inp, max_tokens=steps + 16)
    return ops.argmax(lg[:, -1].contiguous())[:, None], cache
t, cache = prefill(); torch.cuda.synchronize()
res = []
for _ in range(3):
    t0 = time.perf_counter(); t, cache = prefill(); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) * 1e3)
S = ids.shape[1]
print(f"B={B} prefill of {B} x {S} tokens ({17 * B} crops): {min(res):.1f} ms  ({B * S / min(res) * 1e3:.0f} prompt tok/s)")
for _ in range(8): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): lg, t = model.greedy_step(t, cache)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
print(f"B={B} decode: {ms:.3f} ms/step = {B / ms * 1e3:.0f} tok/s per GPU")
