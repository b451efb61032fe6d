"""Where does |HIP - oracle| on the logits come from?  Per-layer decomposition at FULL size.

Two halves, because the oracle needs ~40 GB of host RAM and minutes of CPU, and the HIP path needs the GPU box:

  python tools/precision_decomp.py oracle [c1|c2]     (build container, CPU)
      runs the oracle on the fixture's request in the reference's dtype flow (fp32 attention, phi.py:451-460) and again
      with ONE kind of rounding of the build's attention injected at a time:
          q / k : rotated queries / keys rounded to bf16 (the build stores bf16 K and feeds bf16 MFMA operands)
          p     : softmax probabilities rounded to bf16 (the PV product's MFMA operand)
          o     : attention output rounded to bf16 before o_proj (the build's o_proj is a bf16 GEMM)
          Q K P : the same three roundings to fp16 (11-bit mantissa) -- the candidate fix
      and records, per layer, the relative deviation of the residual stream from the unmodified oracle, plus the
      z-space logit error (the fixture's tolerance unit, tests/golden/gen_golden_oracle.py) under the fixture's lm_head.
      Writes tools/data/precision_<cfg>.npz (per-layer hidden states of the LAST 32 positions of the unmodified oracle,
      git-ignored, travels to the GPU box) and prints the table.

  python tools/precision_decomp.py hip [c1|c2]        (GPU box)
      runs the HIP model on the same request with a hidden-state hook, compares with the oracle's per-layer states.

The table the two halves print is committed as profiles/r03_precision_decomposition.txt.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
DATA = os.path.join(ROOT, "tools", "data")
BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32
KEEP = 32                                  # last positions whose per-layer states are kept


def request(which):
    from golden_inputs import c4_share
    from phi_3_vision_mlx_amd.processor import Phi3VProcessor
    if which == "c1":
        return {"input_ids": np.random.default_rng(0).integers(3, 32000, (1, 128)).astype(np.int64)}
    return c4_share(Phi3VProcessor(None).img_processor)[0]          # config 2 = bench.py's rank-0 request


def fixture_head(which, base):
    from phi_3_vision_mlx_amd.weights import peaked_lm_head
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"{which}_oracle.npz"))
    return peaked_lm_head(base.to(F32), float(fx["spread"][0]), int(fx["head_seed"][0]))


def z_err(lg, ref, head):
    """worst |dlogit_v| / n_v as a fraction of max_u |z_u| (the fixtures' rel_tol unit)."""
    n = head.to(F32).norm(dim=-1).clamp_min(1e-30)
    return (((lg.float() - ref.float()).abs() / n).max() / (ref.float() / n).abs().max()).item()


def oracle_half(which):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import phi3v_oracle as orc
    from phi_3_vision_mlx_amd.config import make_config, phi3v_config_dict
    from phi_3_vision_mlx_amd.weights import synth_weights
    torch.set_num_threads(8)
    cfg = make_config(phi3v_config_dict(vision=True))
    w = synth_weights(cfg, seed=0)
    head = fixture_head(which, w["lm_head.weight"])
    inp = request(which)
    inp = {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}

    def rnd(x, dt):
        return x.to(dt).to(F32)

    def make(flags):
        o = orc.OraclePhi3V(cfg, w, cache_fp32=True)

        def attention(x, i, cache, cos, sin, allowed, n_beam):
            p = f"model.layers.{i}.self_attn."
            nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
            hd = cfg.hidden_size // nh
            B, L, _ = x.shape
            qkv = o.proj(x, p + "qkv_proj.weight")
            q, k, v = torch.split(qkv, [nh * hd, nkv * hd, nkv * hd], dim=-1)
            q, k, v = (t.reshape(B, L, n_, -1).transpose(1, 2) for t, n_ in ((q, nh), (k, nkv), (v, nkv)))
            q, k = orc.rotate_half(q, cos, sin), orc.rotate_half(k, cos, sin)
            if "q" in flags: q = rnd(q, BF16)
            if "Q" in flags: q = rnd(q, F16)
            if "k" in flags: k = rnd(k, BF16)
            if "K" in flags: k = rnd(k, F16)
            k, v = cache(k, v, n_beam)
            s = (q * (hd ** -0.5)) @ k.transpose(-1, -2)
            pr = orc.masked_softmax(s, allowed)
            if "p" in flags: pr = rnd(pr, BF16)
            if "P" in flags: pr = rnd(pr, F16)
            ov = (pr @ v.to(F32)).transpose(1, 2).reshape(B, L, -1)
            if "o" in flags: ov = rnd(ov, BF16)
            return o.proj(ov, p + "o_proj.weight").to(qkv.dtype)
        if flags:
            o.attention = attention
        return o

    def run(flags):
        hs = []
        t0 = time.time()
        o = make(flags)
        x, _ = o.backbone(inp["input_ids"], inp.get("pixel_values"), inp.get("image_sizes"), inp.get("positions"), None,
                          None, None, 4, None, 1, hidden_hook=lambda i, h: hs.append(h[0, -KEEP:].clone()))
        lg = orc._linear(x[:, -1:], head)[0, -1]
        print(f"  oracle[{flags or 'reference flow'}]: {time.time() - t0:.0f} s", flush=True)
        return torch.stack(hs), lg

    base_h, base_lg = run("")
    os.makedirs(DATA, exist_ok=True)
    np.savez(os.path.join(DATA, f"precision_{which}.npz"), hidden_bf16=base_h.to(BF16).view(torch.int16).numpy(),
             logits=base_lg.float().numpy())
    variants = ("q", "k", "p", "o", "qkpo", "QKPo", "QKP") if which == "c1" else ("qkpo", "QKPo")
    rows = []
    for fl in variants:
        h, lg = run(fl)
        d = ((h.float() - base_h.float()).norm(dim=-1) / base_h.float().norm(dim=-1))      # [layers, KEEP]
        rows.append((fl, z_err(lg, base_lg, head), d.mean(1)))
    print(f"\n{which}: oracle with the build's roundings injected vs the oracle in the reference's dtype flow")
    print("variant   z-space logit err   residual-stream deviation |dx|/|x| (mean over the last 32 positions) after layer 0 / 7 / 15 / 23 / 31")
    for fl, ze, d in rows:
        print(f"  {fl:6s}  {100 * ze:6.2f} %          " + "  ".join(f"{100 * d[i].item():.3f} %" for i in (0, 7, 15, 23, 31)))


HEAVY = {"c2h": {}, "c5wh": dict(quantized_fp8=True, use_quantized_cache=True, fp8_activations=False),
         "c5h": dict(quantized_fp8=True, use_quantized_cache=True, fp8_activations=True)}


def hip_half(which):
    """which: c1 | c2 (plain weights) or [tiny_]c2h | c5wh | c5h (heavy-tailed activations, tests/golden/gen_golden_oracle.py
    `heavy`: the oracle side applies the same quantisers)."""
    from phi_3_vision_mlx_amd.api import load_synthetic
    fx = np.load(os.path.join(DATA, f"precision_{which}.npz"))
    ref_h = torch.from_numpy(fx["hidden_bf16"]).view(BF16).float()                          # [layers, KEEP, H]
    tag = which.replace("tiny_", "")
    if tag in HEAVY:
        tiny = which.startswith("tiny_")
        model, proc = load_synthetic(tiny=tiny, seed=0, device="cuda:0", outliers=True, std_scale=4.0 if tiny else 1.0, **HEAVY[tag])
        from golden_inputs import vqa_request
        inp = vqa_request(proc.img_processor, 0)
        hs = []
        model.hidden_hook = lambda i, x, B, L: hs.append(x.view(B, L, -1)[0, -KEEP:].float().cpu())
        model(**inp, max_tokens=4, full_logits=True)
        model.hidden_hook = None
        h = torch.stack(hs)
        d = (h - ref_h).norm(dim=-1) / ref_h.norm(dim=-1)
        ch = __import__("phi_3_vision_mlx_amd.weights", fromlist=["x"]).outlier_channels(model.cfg, 6, 0)
        rest = torch.ones(h.shape[-1], dtype=torch.bool)
        rest[ch] = False
        d_rest = (h - ref_h)[..., rest].norm(dim=-1) / ref_h[..., rest].norm(dim=-1)
        print(f"\n{which}: HIP vs the oracle with the same weights / quantisers, heavy-tailed activations")
        print(f"  outlier channels {ch.tolist()}: |x| there / median |x| = {(ref_h[-1][:, ch].abs().mean() / ref_h[-1][:, rest].abs().median()).item():.0f}x")
        print("  residual-stream deviation |dx|/|x|, all channels, after layer " + "  ".join(f"{i}: {100 * d[i].mean().item():.3f} %" for i in sorted({0, len(d) // 4, len(d) // 2, 3 * len(d) // 4, len(d) - 1})))
        print("  the same over the NON-outlier channels only:               " + "  ".join(f"{i}: {100 * d_rest[i].mean().item():.3f} %" for i in sorted({0, len(d) // 4, len(d) // 2, 3 * len(d) // 4, len(d) - 1})))
        print("  per layer (non-outlier channels): " + " ".join(f"{100 * d_rest[i].mean().item():.2f}" for i in range(d.shape[0])))
        return
    ref_lg = torch.from_numpy(fx["logits"])
    model, _ = load_synthetic(tiny=False, seed=0, device="cuda:0")
    head = fixture_head(which, model.w["lm_head.weight"].cpu())
    model.w["lm_head.weight"] = head.to(BF16).to(model.device)
    inp = request(which)
    hs = []
    model.hidden_hook = lambda i, x, B, L: hs.append(x.view(B, L, -1)[0, -KEEP:].float().cpu())
    logits, _ = model(**inp, max_tokens=4, full_logits=True)
    model.hidden_hook = None
    h = torch.stack(hs)
    d = (h - ref_h).norm(dim=-1) / ref_h.norm(dim=-1)
    print(f"\n{which}: HIP vs the oracle in the reference's dtype flow")
    print(f"  z-space logit err {100 * z_err(logits[0, -1].float().cpu(), ref_lg, head):.2f} %")
    print("  residual-stream deviation |dx|/|x| after layer " + "  ".join(f"{i}: {100 * d[i].mean().item():.3f} %" for i in (0, 7, 15, 23, 31)))
    print("  per layer: " + " ".join(f"{100 * d[i].mean().item():.2f}" for i in range(d.shape[0])))


if __name__ == "__main__":
    half, which = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c1")
    (oracle_half if half == "oracle" else hip_half)(which)
