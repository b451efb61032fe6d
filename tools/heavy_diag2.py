"""Per-layer |HIP - oracle| at every decode step of the tiny heavy-tail c5h / c5wh fixtures (live oracle with the same quantisers)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/tests/golden", ROOT + "/oracle"]
import numpy as np, torch
import phi3v_oracle as orc
import gen_golden_oracle as gg
from test_model_gpu import GOLDEN, HEAVY
from golden_inputs import vqa_request
from phi_3_vision_mlx_amd.api import load_synthetic
from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
from phi_3_vision_mlx_amd.weights import synth_weights
tag = sys.argv[1] if len(sys.argv) > 1 else "c5h"
w8, kv8 = tag in ("c5h", "c5wh", "w8only"), tag in ("c5h", "c5wh", "kv8only")
g = np.load(f"{GOLDEN}/tiny_c5wh_oracle.npz")
kw = dict(quantized_fp8=w8, use_quantized_cache=kv8, fp8_activations=tag == "c5h")
model, proc = load_synthetic(tiny=True, seed=0, device="cuda:0", std_scale=4.0, outliers=True, lm_head_spread=float(g["spread"][0]),
                             lm_head_seed=int(g["head_seed"][0]), **kw)
cfg = make_config(tiny_config_dict(vision=True))
w = synth_weights(cfg, seed=0, std_scale=4.0, outliers=True)
if w8:
    w = gg.c5_quantisers(cfg, w)
o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
if w8:
    o.proj = gg.c5_proj(o, tag == "c5h")
if kv8:
    orc.OracleKVCache = gg.QuantKVCache
inp = vqa_request(proc.img_processor, 0)
oin = {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}
toks = torch.as_tensor(g["tokens"]).long()
oh, hh = [], []
x, ocache = o.backbone(oin["input_ids"], oin["pixel_values"], oin["image_sizes"], oin["positions"], None, None, None, 4, None, 1,
                       hidden_hook=lambda i, h: oh.append(h[0, -1].float().clone()))
model.hidden_hook = lambda i, xx, B, L: hh.append(xx.view(B, L, -1)[0, -1].float().cpu())
inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
logits, cache = model(**inp, max_tokens=4, full_logits=True)
def report(step):
    d = [((a - b).norm() / b.norm()).item() for a, b in zip(hh, oh)]
    print(f"step {step}: per-layer |dx|/|x| (%):", [round(100 * v, 2) for v in d], flush=True)
    oh.clear(), hh.clear()
report(0)
for step in range(1, 4):
    t = toks[:, step - 1:step]
    o.backbone(t, None, None, None, ocache, None, None, 0, None, 1, hidden_hook=lambda i, h: oh.append(h[0, -1].float().clone()))
    model(input_ids=t.to("cuda:0", torch.int32), cache=cache)
    report(step)
    st = cache[0].state
    past = st.offset
    if not kv8:
        continue
    # the row the step appended: dequantised HIP key / value vs the oracle's stored (quantised) copy
    for layer in range(cfg.num_hidden_layers):
        kq = (st.k8[layer, 0, :, past - 1].float() - 128) * st.ks[layer, 0, :, past - 1, None]
        vq = (st.v8[layer, 0, :, :, past - 1].float() - 128) * st.vs[layer, 0, :, past - 1, None]
        ok, ov = ocache[layer].kv[0, 0, :, past - 1], ocache[layer].kv[1, 0, :, past - 1]
        print(f"   layer {layer}: appended K row max|diff| {(kq.cpu() - ok).abs().max():.4f} (|k|max {ok.abs().max():.2f})   V row {(vq.cpu() - ov).abs().max():.4f} (|v|max {ov.abs().max():.2f})")
