"""bench.py's configs[1] request prefilled a few times (tools/prefill_trace.sh runs this under rocprofv3 --kernel-trace and prints the
timeline of the LAST prefill).  argv: [reps] [c1]  (c1: the 128-token text prompt of configs[0])"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from phi_3_vision_mlx_amd import ops
from phi_3_vision_mlx_amd.api import load_synthetic
from phi_3_vision_mlx_amd.workloads import vqa_request

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c1 = len(sys.argv) > 2 and sys.argv[2] == "c1"
model, processor = load_synthetic(blind_model=c1, tiny=False, seed=0, device="cuda:0")
if c1:
    ids = np.random.default_rng(0).integers(3, 32000, size=(1, 128))
    inputs = {"input_ids": ids}
else:
    vqa_request(processor.img_processor, 0, device="cuda:0")
    inputs = vqa_request(processor.img_processor, 0, device="cuda:0")
torch.cuda.synchronize()
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(100)                        # at::cuda spin_kernel: the trace splits prefills at it
    logits, cache = model(**inputs, max_tokens=72)
    tok = ops.argmax(logits[:, -1, :].contiguous())[:, None].tolist()
    print(f"prefill {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms, first token {tok}", flush=True)
    del cache
