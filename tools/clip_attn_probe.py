"""CLIP-shaped attention (17 crops x 16 heads x 577 tokens x 64, non-causal): every prompt-sized kernel on the same data, timed
alternately, each checked against an fp32 torch reference.  Run on the GPU box: python tools/clip_attn_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phi_3_vision_mlx_amd import ops
from tools.bench_kernels import timeit

B, L, nh, hd = 17, 577, 16, 64
Tp = (L + 63) // 64 * 64
torch.manual_seed(0)
qf = torch.randn(B, nh, L, hd, device="cuda")
k = torch.zeros(B, nh, Tp, hd, device="cuda", dtype=torch.bfloat16)
v = torch.zeros(B, nh, hd, Tp, device="cuda", dtype=torch.bfloat16)
k[:, :, :L] = torch.randn(B, nh, L, hd, device="cuda").bfloat16()
v[:, :, :, :L] = torch.randn(B, nh, hd, L, device="cuda").bfloat16()
scale = hd ** -0.5
q_plain, q_pre = qf.bfloat16(), (qf * scale * ops.Q_PRESCALE).bfloat16()
ref = torch.softmax((q_plain.float() * scale) @ k[:, :, :L].float().transpose(-1, -2), -1) @ v[:, :, :, :L].float().transpose(-1, -2)
ref = ref.transpose(1, 2).reshape(B, L, nh * hd)
out = torch.empty(B, L, nh * hd, device="cuda", dtype=torch.bfloat16)
fl = 4 * B * nh * L * L * hd
kinds = [("dma plain", q_plain, False, 0, 0, -1), ("dma prescaled", q_pre, True, 0, 0, -1), ("pingpong prescaled", q_pre, True, 1, 0, -1),
         ("interleaved 4 waves", q_pre, True, 1, 1, 4), ("interleaved 8 waves", q_pre, True, 1, 1, 8)]
t = {n: [] for n, *_ in kinds}
for rep in range(3):
    for n, q, pre, pp, il, nw in kinds:
        olds = ops.set_tuning("attn_pp", pp), ops.set_tuning("attn_il", il), ops.set_tuning("attn_il_waves", nw)
        fn = lambda i: ops.attention(q, out, B, L, nh, nh, hd, scale, False, k_past=k, v_past=v, past_t=Tp, new_is_cache=True, q_prescaled=pre)
        t[n].append(timeit(fn, 1, iters=20))
        if rep == 0:
            err = (out.float() - ref).abs().max().item()
            print(f"{n:22s} max |err| vs fp32 reference {err:.4f} (|ref| max {ref.abs().max().item():.2f})", flush=True)
        ops.set_tuning("attn_pp", olds[0]), ops.set_tuning("attn_il", olds[1]), ops.set_tuning("attn_il_waves", olds[2])
for n, *_ in kinds:
    ms = sorted(t[n])[1]
    print(f"clip attention {n:22s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TF/s")
