"""Layer-0 projections of ONE decode token under heavy tails: HIP (M = 1 paths) vs the oracle's arithmetic, stage by stage."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/tests/golden", ROOT + "/oracle"]
import numpy as np, torch
import phi3v_oracle as orc
import gen_golden_oracle as gg
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
tiny = sys.argv[1] == "tiny"
tok = int(sys.argv[2])
model, proc = load_synthetic(blind_model=True, tiny=tiny, seed=0, device="cuda:0", std_scale=4.0 if tiny else 1.0, outliers=True,
                             quantized_fp8=True)
cfg = model.cfg
BF16, F32 = torch.bfloat16, torch.float32
def rel(a, b): return ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
emb = model.w["model.embed_tokens.weight"][tok:tok + 1].contiguous()               # [1, H] bf16
x_o = emb.cpu()
p = "model.layers.0."
def W(key):                                                                          # the oracle's weight: e4m3 x scale in fp32
    w8, sc = model.w8[key]
    return (w8.view(torch.float8_e4m3fn).float() * sc[:, None]).cpu()
n1 = model.w[p + "input_layernorm.weight"]
qkv_h = model._proj(emb, p + "self_attn.qkv_proj.weight", norm_w=n1)
xn = orc.rms_norm(x_o, n1.cpu(), cfg.rms_norm_eps)
qkv_o = (xn.float() @ W(p + "self_attn.qkv_proj.weight").t()).to(BF16)
print(f"token {tok}: |x|max {x_o.float().abs().max():.1f} rms {x_o.float().pow(2).mean().sqrt():.2f}; normalised |xn|max {xn.float().abs().max():.1f}")
print(f"  qkv (fused norm + fp8 weights, M = 1): rel diff {100 * rel(qkv_h, qkv_o):.3f} %   max|qkv| {qkv_o.float().abs().max():.1f}")
# attention of a single token over itself = V (softmax weight 1): o = v
H, nh = cfg.hidden_size, cfg.num_attention_heads
v_o = qkv_o[:, -H:] if cfg.num_key_value_heads == nh else None
o_h = model._proj(v_o.to("cuda:0").contiguous(), p + "self_attn.o_proj.weight", ops.EPI_RESID_BF16, resid=emb.clone())
o_o = (x_o.float() + (v_o.float() @ W(p + "self_attn.o_proj.weight").t()).to(BF16).float()).to(BF16)
print(f"  o_proj + residual (same input on both sides): rel diff {100 * rel(o_h, o_o):.3f} %")
n2 = model.w[p + "post_attention_layernorm.weight"]
a_h = model._proj(o_o.to("cuda:0").contiguous(), p + "mlp.gate_up_proj.weight", ops.EPI_SILU_MUL, norm_w=n2)
hn = orc.rms_norm(o_o, n2.cpu(), cfg.rms_norm_eps)
y = (hn.float() @ W(p + "mlp.gate_up_proj.weight").t()).to(BF16)
gate, up = torch.chunk(y, 2, dim=-1)
a_o = (gate * torch.sigmoid(gate)) * up
print(f"  gate_up + SiLU*up (same input): rel diff {100 * rel(a_h, a_o):.3f} %   max|gate| {gate.float().abs().max():.1f} max|act*up| {a_o.float().abs().max():.1f}")
bad = ((a_h.float().cpu() - a_o.float()).abs() / a_o.float().abs().max()).topk(3)
print("     worst entries:", [(int(i), float(a_h[0, i]), float(a_o[0, i]), float(gate[0, i]), float(up[0, i])) for i in bad.indices[0]])
d_h = model._proj(a_o.to("cuda:0").contiguous(), p + "mlp.down_proj.weight", ops.EPI_RESID_BF16, resid=o_o.to("cuda:0").clone())
d_o = (o_o.float() + (a_o.float() @ W(p + "mlp.down_proj.weight").t()).to(BF16).float()).to(BF16)
print(f"  down_proj + residual (same input): rel diff {100 * rel(d_h, d_o):.3f} %")
