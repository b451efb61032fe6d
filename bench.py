#!/usr/bin/env python3
"""Headline benchmark: Phi-3-Vision bf16 single-image VQA on MI355X
(BASELINE.json `configs[1]`): prefill ms + decode tokens/s, with the roofline
of the dominant decode kernel and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU, each with its own replica and its own request (batch
sharding, no data-path collective -> "weak" scaling); RCCL is used for the
barriers, the max-over-ranks of the timed span and the token gather.
A "step" = one greedy decode step (one new token per sequence) through the
same graph-replayed path `generate()` uses; timing definitions follow the
reference (phi_3_vision_mlx.py:384-403): prefill = first `model(**inputs)` +
argmax + sync, timer started AFTER host preprocessing.
Synthetic data: seeded random weights of the real architecture, one seeded
random 336x336 image (-> 1344x1344 HD, 17 crops, 2509 image tokens) + 20
random text tokens.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--prefill-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="tiny config (debug only; the result is NOT the headline metric)")
    return ap.parse_args()


def usable_cores(cap=64):
    """Threads the CPU leg may really use: affinity mask and cgroup CPU quota, capped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline(model, ctx_len, n_layers_sample=4, n_tokens=2):
    """Oracle (CPU restatement of phi.py) timed on the host cores, bounded sample:
    `n_tokens` greedy decode steps at the same context length through
    `n_layers_sample` of the decoder layers (fp32 weights resident) + final norm + lm_head,
    extrapolated linearly to all layers.  Baseline only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import phi3v_oracle as orc
    cfg = model.cfg
    cores = usable_cores()
    torch.set_num_threads(cores)
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    for i in range(n_layers_sample):
        names += [k for k in model.w if k.startswith(f"model.layers.{i}.")]
    w = {k: model.w[k].cpu() for k in names}
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    nkv, hd = cfg.num_key_value_heads, cfg.hidden_size // cfg.num_attention_heads
    caches = []
    for i in range(n_layers_sample):
        c = orc.OracleKVCache(cfg, 1, ctx_len, n_tokens + 1)
        c.kv = torch.randn(c.shape, dtype=torch.float32) * 0.5
        c.offset = ctx_len
        caches.append(c)
    cos, sin = orc.su_rope_tables(cfg, ctx_len + n_tokens + 1, None)
    tok = torch.tensor([[17]])

    def step(t):
        x = o.embed(tok)
        past = caches[0].offset
        allowed = torch.ones((1, 1, 1, past + 1), dtype=torch.bool)
        for i in range(n_layers_sample):
            x = o.decoder_layer(x, i, caches[i], cos[:, :, past:past + 1], sin[:, :, past:past + 1], allowed, 1)
        t0 = time.perf_counter()
        lg = orc._linear(orc.rms_norm(x, o.W("model.norm.weight"), cfg.rms_norm_eps), o.W("lm_head.weight"))
        return lg, time.perf_counter() - t0
    step(0)                                                     # warm-up (fp32 weight copies, page-in)
    for c in caches:
        c.offset = ctx_len
    t_layers = t_head = 0.0
    for t in range(n_tokens):
        t0 = time.perf_counter()
        _, th = step(t)
        dt = time.perf_counter() - t0
        t_head += th
        t_layers += dt - th
    per_tok = (t_layers / n_tokens) * (cfg.num_hidden_layers / n_layers_sample) + t_head / n_tokens
    return {"value": round(1.0 / per_tok, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n_tokens} decode steps at context {ctx_len}, {n_layers_sample}/{cfg.num_hidden_layers} decoder layers "
                      f"+ lm_head on torch-CPU fp32, extrapolated to all layers; prefill not timed on CPU"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    share = bool(os.environ.get("P3V_BENCH_SHARE_GPU"))           # debug only: all ranks on GPU 0 over gloo (1-GPU boxes)
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    from PIL import Image

    t0 = time.perf_counter()
    model, processor = load_synthetic(blind_model=False, tiny=args.tiny, seed=0, device=dev)
    torch.cuda.synchronize()
    t_weights = time.perf_counter() - t0
    cfg = model.cfg

    # ---- request: one 336x336 image (seed = rank) + 20 random text tokens
    rng = np.random.default_rng(rank)
    img = Image.fromarray(rng.integers(0, 256, (336, 336, 3), dtype=np.uint8))
    t0 = time.perf_counter()
    # image stage exactly as `processor(text, images)` runs it: resize / pad / normalise / crop on the GPU (same bits as the
    # host path, P3V_HOST_PREPROCESS=1 selects that one), outside the prefill timer like the reference's processor
    if os.environ.get("P3V_HOST_PREPROCESS") == "1":
        image_inputs = processor.img_processor([img], dtype=np.float32)
        pixel_values = processor._to_device(image_inputs["pixel_values"])
    else:
        processor.img_processor.device_call([img], dev)            # warm-up (first launch of the kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        image_inputs = processor.img_processor.device_call([img], dev)
        pixel_values = image_inputs["pixel_values"]
    torch.cuda.synchronize()
    host_pre_ms = (time.perf_counter() - t0) * 1e3
    n_img = image_inputs["num_img_tokens"][0]
    text_ids = rng.integers(3, 32000, 20)
    ids = np.concatenate([[1], text_ids[:8], -np.ones(n_img, dtype=np.int64), [1], text_ids[8:]])[None].astype(np.int64)
    inputs = {"input_ids": ids, "pixel_values": pixel_values,
              "image_sizes": np.asarray(image_inputs["image_sizes"]), "positions": np.argwhere(ids < 0)}
    S = ids.shape[1]
    max_tokens = args.warmup + args.steps + 8

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- prefill (ViT + projector + decoder prefill + lm_head + argmax + sync)
    prefill_ms = []
    for rep in range(args.prefill_reps + 1):
        barrier()
        t0 = time.perf_counter()
        logits, cache = model(**inputs, max_tokens=max_tokens)
        token = ops.argmax(logits[:, -1, :].contiguous())[:, None]
        first = token.tolist()
        dt = (time.perf_counter() - t0) * 1e3
        if rep > 0:
            prefill_ms.append(dt)
    prefill = float(np.median(prefill_ms))
    print("prefill reps ms:", [round(v, 1) for v in prefill_ms], file=sys.stderr)

    # ---- decode: W untimed + K timed graph-replayed greedy steps
    for _ in range(args.warmup):
        logits, token = model.greedy_step(token, cache)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits, token = model.greedy_step(token, cache)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        pf = torch.tensor([prefill], device=dev)
        dist.all_reduce(pf, op=dist.ReduceOp.MAX)
        prefill = pf.item()
        hist = cache[0].state.graphs["greedy"]["history"]
        gathered = [torch.empty_like(hist) for _ in range(world)] if rank == 0 else None
        dist.gather(hist, gathered, dst=0)                        # token gather over RCCL (request boundary only)
    B = 1
    tokens_per_s = world * B * args.steps / elapsed

    # ---- roofline of the dominant decode kernel (gate_up GEMV + fused RMSNorm + SiLU*up): the 32 layers' launches
    #      replayed as one hipGraph on the launch stream (exactly the launches of a decode step, weights of every
    #      layer in turn so nothing is cache-resident), HIP events around the replay
    I, H, L = cfg.intermediate_size, cfg.hidden_size, cfg.num_hidden_layers
    alg_bytes = 2 * I * H * 2                                     # bf16 [2I, H] weights streamed once per launch
    xb = torch.randn((1, H), device=dev).to(torch.bfloat16)
    ab = torch.empty((1, I), dtype=torch.bfloat16, device=dev)

    def gate_up_all_layers():
        for i in range(L):
            ops.gemv(xb, model.w[f"model.layers.{i}.mlp.gate_up_proj.weight"], ops.EPI_SILU_MUL,
                     norm_w=model.w[f"model.layers.{i}.post_attention_layernorm.weight"], norm_eps=cfg.rms_norm_eps, out=ab)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gate_up_all_layers()
        gr = ops.Graph()
        gr.begin()
        gate_up_all_layers()
        gr.end()
        gr.launch()
        side.synchronize()
        reps = []
        for _ in range(5):
            e0, e1 = ops.Event(), ops.Event()
            e0.record()
            gr.launch()
            e1.record()
            side.synchronize()
            reps.append(e0.elapsed_ms(e1) / L)
    k_ms = float(np.median(reps))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    # HBM bytes per launch from the PMC passes (tools/pmc_gemv.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # runs, FETCH_SIZE x2 on gfx950); a committed measurement, not re-collected inside the timed benchmark
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_gate_up_gemv.json")) as f:
            traffic = json.load(f)["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    kv_bytes = 2 * L * cfg.num_key_value_heads * (cfg.hidden_size // cfg.num_attention_heads) * 2 * (S + args.warmup + args.steps // 2)
    w_bytes = sum(v.numel() * 2 for k, v in model.w.items() if k.startswith("model.layers.") or k == "lm_head.weight")
    step_s = elapsed / args.steps
    out = {
        "metric": "decode tokens/sec (+ prefill ms), Phi-3-Vision bf16 1-image VQA",
        "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Phi-3-Vision single 336x336 image VQA (BASELINE configs[1]); 17 crops, "
                               f"{n_img} image tokens + 22 text tokens, prompt {S}, B=1 per GPU, greedy, EOS suppressed",
                   "parallelism": f"batch-sharded replicas x{world}", "tiny": bool(args.tiny)},
        "prefill_ms": round(prefill, 3), "prefill_tokens": int(S), "preprocess_ms": round(host_pre_ms, 1), "preprocess": "host" if os.environ.get("P3V_HOST_PREPROCESS") == "1" else "device",
        "decode_step_hbm": {"algorithmic_GB_per_token": round((w_bytes + kv_bytes) / 1e9, 3),
                            "achieved_GBps": round((w_bytes + kv_bytes) / step_s / 1e9, 1),
                            "frac_of_peak": round((w_bytes + kv_bytes) / step_s / 1e9 / HBM_PEAK_GBS, 4)},
        "roofline": {"bound": "hbm", "kernel": "k_gemv3<1,1,6> (RMSNorm + gate_up_proj + SiLU*up), 32 launches/step", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(k_ms, 5)},
        "first_token": first, "weights_init_s": round(t_weights, 1),
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not args.tiny:
        out["cpu_baseline"] = cpu_baseline(model, S)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
