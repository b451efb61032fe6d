"""Golden vectors from the CPU oracle (oracle/phi3v_oracle.py) on seeded synthetic weights:

  tiny_oracle.npz   tiny text + vision models: greedy tokens, per-step top-k logits, choose/constrain results
  c1_oracle.npz     FULL-SIZE Phi-3-mini-128K (BASELINE config 1: 128-token prompt, text-only, greedy):
                    prefill + 7 decode steps; top-16 logits per step.  (~6 min, ~25 GB RAM, run once here.)

  c2_oracle.npz     FULL-SIZE Phi-3-Vision (BASELINE config 2 = bench.py's rank-0 request: one seeded 336x336 image,
                    2531-token prompt, greedy): prefill + 3 decode steps; top-16 logits per step.  (CPU, run once here.)

    python tests/golden/gen_golden_oracle.py [tiny|c1|c2|all]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)
import phi3v_oracle as orc  # noqa: E402
from golden_inputs import make_image  # noqa: E402
from phi_3_vision_mlx_amd.config import make_config, phi3v_config_dict, tiny_config_dict  # noqa: E402
from phi_3_vision_mlx_amd.processor import Phi3FProcessor, Phi3VProcessor  # noqa: E402
from phi_3_vision_mlx_amd.weights import synth_weights  # noqa: E402

TOPK = 16


def topk_pack(lg):
    v, i = lg.float().topk(TOPK, dim=-1)
    return v.numpy().astype(np.float32), i.numpy().astype(np.int32)


def greedy_record(o, inputs, n):
    toks, lgs = orc.greedy_generate(o, dict(inputs), n, stop_on_eos=False)
    v, i = topk_pack(lgs)
    return toks.numpy().astype(np.int32), v, i


def tiny():
    out = {}
    for blind in (True, False):
        tag = "text" if blind else "vis"
        cfg = make_config(tiny_config_dict(vision=not blind))
        w = synth_weights(cfg, seed=0, std_scale=4.0)
        o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
        proc = (Phi3FProcessor if blind else Phi3VProcessor)(None)
        if blind:
            ids = np.random.default_rng(11).integers(3, 32000, (1, 40)).astype(np.int64)
            out["text_ids"] = ids
            t, v, i = greedy_record(o, {"input_ids": ids}, 8)
            out["text_tokens"], out["text_topv"], out["text_topi"] = t, v, i
            prompts = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
            inp = proc(prompts)
            t, v, i = greedy_record(o, inp, 6)
            out["batch_tokens"], out["batch_topv"], out["batch_topi"] = t, v, i
            opts = proc([f" {c}" for c in "ABCDE"])["input_ids"][:, -1]
            out["choose_idx"] = np.asarray(orc.choose_from(o, proc(prompts), opts), dtype=np.int32)
            idc = proc.tokenizer.encode(" The answer is", add_special_tokens=False)[1:]
            for ub in (False, True):
                s, sc = orc.constrain_one(o, dict(inp), (4, " The answer is"), idc, use_beam=ub)
                out[f"constrain_beam{int(ub)}_synth"] = s.numpy().astype(np.int32)
                out[f"constrain_beam{int(ub)}_score"] = sc.float().numpy()
        else:
            inp = proc("<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n", [make_image(336, 336, "noise", 0)])
            t, v, i = greedy_record(o, inp, 4)
            out["vis_tokens"], out["vis_topv"], out["vis_topi"] = t, v, i
            out["vis_n_ids"] = np.asarray([np.asarray(inp["input_ids"]).shape[1]], dtype=np.int32)
        print(tag, "done")
    np.savez_compressed(os.path.join(HERE, "tiny_oracle.npz"), **out)
    print("wrote tiny_oracle.npz")


def c1():
    torch.set_num_threads(8)
    cfg = make_config(phi3v_config_dict(vision=False))
    t0 = time.time()
    w = synth_weights(cfg, seed=0)
    print(f"weights {time.time()-t0:.0f}s")
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    ids = np.random.default_rng(0).integers(3, 32000, (1, 128)).astype(np.int64)
    t0 = time.time()
    toks, lgs = orc.greedy_generate(o, {"input_ids": ids}, 8, stop_on_eos=False)
    print(f"prefill + 7 decode steps {time.time()-t0:.0f}s, tokens {toks.tolist()}")
    v, i = topk_pack(lgs)
    np.savez_compressed(os.path.join(HERE, "c1_oracle.npz"), ids=ids, tokens=toks.numpy().astype(np.int32), topv=v, topi=i,
                        absmax=lgs.float().abs().amax(dim=-1).numpy())
    print("wrote c1_oracle.npz")


def c2_request(img_processor):
    """bench.py's rank-0 request, rebuilt here so that the fixture pins the benchmarked workload."""
    from PIL import Image
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (336, 336, 3), dtype=np.uint8))
    image_inputs = img_processor([img])
    n_img = image_inputs["num_img_tokens"][0]
    text_ids = rng.integers(3, 32000, 20)
    ids = np.concatenate([[1], text_ids[:8], -np.ones(n_img, dtype=np.int64), [1], text_ids[8:]])[None].astype(np.int64)
    return {"input_ids": ids, "pixel_values": np.asarray(image_inputs["pixel_values"], dtype=np.float32),
            "image_sizes": np.asarray(image_inputs["image_sizes"]), "positions": np.argwhere(ids < 0)}


def c2():
    torch.set_num_threads(8)
    cfg = make_config(phi3v_config_dict(vision=True))
    t0 = time.time()
    w = synth_weights(cfg, seed=0)
    print(f"weights {time.time()-t0:.0f}s", flush=True)
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    inp = c2_request(Phi3VProcessor(None).img_processor)
    t0 = time.time()
    toks, lgs = orc.greedy_generate(o, {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}, 4,
                                    stop_on_eos=False)
    print(f"prefill + 3 decode steps {time.time()-t0:.0f}s, tokens {toks.tolist()}", flush=True)
    v, i = topk_pack(lgs)
    np.savez_compressed(os.path.join(HERE, "c2_oracle.npz"), n_ids=np.asarray([inp["input_ids"].shape[1]], dtype=np.int32),
                        tokens=toks.numpy().astype(np.int32), topv=v, topi=i, absmax=lgs.float().abs().amax(dim=-1).numpy())
    print("wrote c2_oracle.npz")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("tiny", "all"):
        tiny()
    if which in ("c1", "all"):
        c1()
    if which in ("c2", "all"):
        c2()
