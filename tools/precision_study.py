"""Which bf16 rounding in the attention block costs the most logit accuracy vs the reference's fp32 flow?
CPU-only study on the oracle (8-layer, H=768 config with head_dim 96)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import phi3v_oracle as orc
from phi_3_vision_mlx_amd.config import make_config, tiny_config_dict
from phi_3_vision_mlx_amd.weights import synth_weights
d = tiny_config_dict(vision=False); d.update(hidden_size=768, num_attention_heads=8, num_key_value_heads=8, num_hidden_layers=int(sys.argv[1]) if len(sys.argv) > 1 else 8, intermediate_size=2048)
cfg = make_config(d)
w = synth_weights(cfg, seed=0, std_scale=1.5)
ids = np.random.default_rng(0).integers(3, 32000, (1, 128))
BF16, F32 = torch.bfloat16, torch.float32
def rb(x): return x.to(BF16).to(F32)
def make(flags):
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    def attention(x, i, cache, cos, sin, allowed, n_beam):
        p = f"model.layers.{i}.self_attn."
        nh, hd = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
        B, L, _ = x.shape
        qkv = orc._linear(x, o.W(p + "qkv_proj.weight"))
        q, k, v = torch.split(qkv, [nh * hd] * 3, dim=-1)
        q, k, v = (t.reshape(B, L, nh, -1).transpose(1, 2) for t in (q, k, v))
        q, k = orc.rotate_half(q, cos, sin), orc.rotate_half(k, cos, sin)
        if "q" in flags: q = rb(q)
        if "k" in flags: k = rb(k)
        k, v = cache(k, v, n_beam)
        s = (q * (hd ** -0.5)) @ k.transpose(-1, -2)
        pr = orc.masked_softmax(s, allowed)
        if "p" in flags: pr = rb(pr)
        ov = (pr @ v.to(F32)).transpose(1, 2).reshape(B, L, -1)
        if "o" in flags: ov = rb(ov)
        return orc._linear(ov, o.W(p + "o_proj.weight")).to(qkv.dtype)
    o.attention = attention
    return o
base, _ = make("")(input_ids=ids, max_tokens=1)
base = base[:, -1].float(); scale = base.abs().max().item()
print(f"layers {cfg.num_hidden_layers}  |logit|max {scale:.2f}  top2 margin {(base.topk(2).values[0,0]-base.topk(2).values[0,1]).item():.3f}")
for flags in ("q", "k", "p", "o", "qk", "qkp", "qkpo"):
    lg, _ = make(flags)(input_ids=ids, max_tokens=1)
    e = (lg[:, -1].float() - base).abs()
    print(f"  round {flags:5s}: max err {e.max().item():.4f} ({e.max().item()/scale*100:.2f}% of max)  mean {e.mean().item():.4f}")
