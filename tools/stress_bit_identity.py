"""Loop the kernel tests that claim BIT identity between a fused launch and the launches it replaces; their rotation tables are drawn
unseeded, so every pass is a new input.  python tools/stress_bit_identity.py [passes]"""
import inspect
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from phi_3_vision_mlx_amd import ops
import test_kernels_gpu as tk

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
names = ["test_q8_attention_decode_with_fused_fp8_oproj_is_bit_identical_to_two_launches", "test_gemm_qkv_fused",
         "test_gemm_resid_norm_is_the_two_calls", "test_attention_decode_with_fused_oproj_is_bit_identical_to_two_launches"]
for name in names:
    fn = getattr(tk, name)
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    argnames = [a.strip() for a in marks[0].args[0].split(",")]
    cases = marks[0].args[1]
    bad = 0
    for p in range(passes):
        torch.manual_seed(1000 + p)
        for c in cases:
            c = c if isinstance(c, (tuple, list)) else (c,)
            try:
                fn(ops, **dict(zip(argnames, c)))
            except AssertionError as e:
                bad += 1
                print(f"{name}{tuple(c)} pass {p}: {str(e)[:200]}", flush=True)
    print(f"{name}: {bad} failures in {passes} passes x {len(cases)} cases", flush=True)
