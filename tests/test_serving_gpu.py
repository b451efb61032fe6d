"""Serving and sharding paths against ORACLE fixtures: the
continuous-batching engine, the HTTP handler on top of it, dist.prefill_requests and dist.generate_sharded produce, per
request, the tokens of the request's own B = 1 oracle run -- the reference runs image prompts at B = 1 only
(phi_3_vision_mlx.py:377-378) and its server one generate per request (server.py:17), so that run IS the reference
behaviour of every request whatever it is batched with.

Comparison rule (the fixtures', tests/golden/gen_golden_oracle.py): a free-running request is compared token by token
up to its first step whose oracle top-2 margin is not clear under the fixture's tolerance; the fixtures' head seeds make
the first step(s) of EVERY request clear, so every request is pinned on at least its first token."""
import base64
import json
import threading
import urllib.request
from io import BytesIO

import numpy as np
import pytest
import torch

from test_model_gpu import GOLDEN, _full_model

pytestmark = pytest.mark.gpu


def tokens_vs_fixture(got, g, what, rows=None, min_first=1, budgets=None):
    """got[i]: EOS-trimmed token list of request rows[i] (budgets[i]: its max_tokens, default = the fixture's steps).
    Returns the number of tokens compared (all equal)."""
    ref, clear = g["tokens"], g["margins"] > 1.0
    rows = range(len(got)) if rows is None else rows
    n = 0
    for k, (toks, r) in enumerate(zip(got, rows)):
        assert clear[r, :min_first].all(), f"{what}: fixture request {r} is not clear on its first step(s)"
        budget = ref.shape[1] if budgets is None else budgets[k]
        assert len(toks) <= budget, f"{what}: request {r} produced {len(toks)} tokens for a budget of {budget}"
        for step in range(min(ref.shape[1], budget)):
            if not clear[r, step]:
                break
            assert step < len(toks), f"{what}: request {r} stopped after {len(toks)} tokens, oracle continues {ref[r].tolist()}"
            assert toks[step] == int(ref[r, step]), f"{what}: request {r} step {step}: {toks} != oracle {ref[r].tolist()}"
            n += 1
            if toks[step] == 32007:
                break
    return n


def _tiny_serve():
    from golden_inputs import serve_requests
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(GOLDEN + "/tiny_serve_oracle.npz")
    model, proc = load_synthetic(blind_model=False, tiny=True, seed=0, std_scale=4.0, device="cuda:0",
                                 lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]))
    reqs = serve_requests(proc)
    assert [int(np.asarray(r["input_ids"]).shape[-1]) for r in reqs] == g["n_ids"].tolist()
    return g, model, proc, reqs


class IdTokenizer:
    """Delegates encoding to the real tokenizer; decodes to the ids themselves so HTTP responses can be compared as tokens."""

    def __init__(self, real):
        self.real = real

    def __call__(self, *a, **kw):
        return self.real(*a, **kw)

    def encode(self, *a, **kw):
        return self.real.encode(*a, **kw)

    def decode(self, ids, **kw):
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, seqs, **kw):
        return [self.decode(s) for s in seqs]


def _png_uri(img):
    buf = BytesIO()
    img.save(buf, format="PNG")
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()


def _post(port, body, timeout=300):
    req = urllib.request.Request(f"http://127.0.0.1:{port}/v1/completions", data=json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return json.loads(r.read())


def test_engine_tiny_requests_join_mid_flight_and_match_the_oracle():
    """engine.ContinuousEngine, 3 slots, 7 requests (one image request of 2534 tokens, texts of 9..115 tokens): four arrive
    first, three while rows are generating; every request's tokens == its own B = 1 ORACLE run."""
    from golden_inputs import SERVE_STEPS
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    g, model, proc, reqs = _tiny_serve()
    eng = ContinuousEngine(model, proc, slots=3, window=4096)
    budgets = [SERVE_STEPS, 3, SERVE_STEPS, SERVE_STEPS, SERVE_STEPS, 5, SERVE_STEPS]    # request 1 leaves early: its row is refilled mid-flight, and so on
    handles = [eng.submit(r, b) for r, b in zip(reqs[:4], budgets)]
    for _ in range(2):
        eng.step()
    handles += [eng.submit(r, b) for r, b in zip(reqs[4:], budgets[4:])]       # arrive while rows are generating
    eng.run_until_idle()
    assert all(h.done.is_set() and h.error is None for h in handles)
    assert eng.joined_mid_flight >= 2 and eng.steps < sum(budgets) and eng.failures == 0     # rows really shared steps
    n = tokens_vs_fixture([h.tokens for h in handles], g, "engine", min_first=2, budgets=budgets)
    assert n >= 16, n
    big = eng.submit(reqs[1], 5000)                                 # beyond the window: refused, not queued
    assert big.done.is_set() and isinstance(big.error, ValueError)
    print(f"tiny engine: {n} free-running tokens of 7 requests equal the oracle's")


def test_prefill_requests_groups_nearly_equal_lengths_and_matches_the_oracle():
    """dist.prefill_requests pads SHORT prompts of different lengths into one prefill group (a short prompt alone still
    streams every weight once): the image request stays its own group, the six text prompts (9..115 tokens) share one; the
    decode batch's tokens == each request's B = 1 oracle run."""
    from golden_inputs import SERVE_STEPS
    from phi_3_vision_mlx_amd import dist
    g, model, proc, reqs = _tiny_serve()
    calls = []
    real = model.prefill_slot

    def spy(st, row, inputs, **kw):
        calls.append(np.asarray(inputs["input_ids"]).reshape(-1, np.asarray(inputs["input_ids"]).shape[-1]).shape[0])
        return real(st, row, inputs, **kw)
    model.prefill_slot = spy
    try:
        got = dist.generate_requests(model, proc, reqs, SERVE_STEPS, return_tokens=True)
    finally:
        model.prefill_slot = real
    assert sorted(calls) == [1, 6], calls
    assert tokens_vs_fixture(got, g, "prefill_requests", min_first=2) >= 20


def test_http_handler_on_the_engine_matches_the_oracle():
    """POST /v1/completions (concurrent clients; request 0 carries its image as a data URI) -> handler -> chat template ->
    continuous-batching engine: the response of every request == its B = 1 oracle tokens."""
    from golden_inputs import SERVE_PROMPTS, SERVE_STEPS, make_image
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    from phi_3_vision_mlx_amd.server import serve_continuous
    g, model, proc, _ = _tiny_serve()
    proc.tokenizer = IdTokenizer(proc.tokenizer)
    httpd, backend = serve_continuous(ContinuousEngine(model, proc, slots=2, window=4096), port=0)
    threading.Thread(target=httpd.serve_forever, daemon=True).start()
    port, got = httpd.server_address[1], {}
    bodies = [{"prompt": SERVE_PROMPTS[0], "images": [_png_uri(make_image(336, 336, "noise", 0))], "max_tokens": SERVE_STEPS},
              {"prompt": SERVE_PROMPTS[1:4], "max_tokens": SERVE_STEPS}, {"prompt": SERVE_PROMPTS[4], "max_tokens": SERVE_STEPS},
              {"prompt": SERVE_PROMPTS[5:7], "max_tokens": SERVE_STEPS}]
    try:
        ths = [threading.Thread(target=lambda i=i, b=b: got.__setitem__(i, _post(port, b))) for i, b in enumerate(bodies)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    finally:
        httpd.shutdown()
        backend.close()
    assert all(got[i]["model"] == "phi-3-vision" for i in range(4))
    texts = got[0]["responses"] + got[1]["responses"] + got[2]["responses"] + got[3]["responses"]
    assert len(texts) == 7
    toks = [[int(t) for t in s.split()] for s in texts]
    assert tokens_vs_fixture(toks, g, "HTTP", min_first=2) >= 20


def _c4_inputs(proc):
    from golden_inputs import c4_share
    share = c4_share(proc.img_processor)
    return [dict(r, pixel_values=torch.from_numpy(r["pixel_values"]).to("cuda:0")) if "pixel_values" in r else r for r in share]


class C4Processor:
    """What the HTTP path needs from a processor, for BASELINE config 4's SYNTHETIC requests (random token ids have no
    text form): the prompt "c4:<i>" names request i of the share; an image request is rebuilt FROM THE IMAGE THE CLIENT
    SENT (decoded data URI -> the real image processor), so the picture really travels through the handler."""

    def __init__(self, proc, share):
        self.proc, self.share, self.tokenizer = proc, share, IdTokenizer(proc.tokenizer)

    def __call__(self, text, images=None):
        i = int(text.split("c4:")[1].split("<")[0])
        r = self.share[i]
        if images is None:
            assert "pixel_values" not in r
            return r
        out = self.proc.img_processor.device_call(images, "cuda:0")
        assert torch.equal(out["pixel_values"].cpu().float(), r["pixel_values"].cpu().float())    # the PNG round trip is lossless
        return dict(r, pixel_values=out["pixel_values"])


def test_c4_share_through_the_engine_and_the_http_handler_full_size():
    """BASELINE config 4's share of one GPU at FULL size (4 single-image VQA requests of 2531 tokens + 4 text prompts of
    65..233 tokens) through engine.ContinuousEngine with 3 slots -- staggered arrivals, every request after the third joins
    while other rows are generating -- and then through the HTTP handler on a fresh engine: every (request, step) token ==
    the per-request B = 1 oracle run of c4_oracle.npz wherever the fixture's margin is clear, the first token always."""
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    from phi_3_vision_mlx_amd.server import serve_continuous
    g = np.load(GOLDEN + "/c4_oracle.npz")
    model, proc = _full_model(g)
    reqs = _c4_inputs(proc)
    assert [r["input_ids"].shape[1] for r in reqs] == g["n_ids"].tolist()
    n_steps = g["tokens"].shape[1]
    eng = ContinuousEngine(model, proc, slots=3, window=4096)
    order = [0, 4, 1, 5, 2, 6, 3, 7]                                # image and text requests interleaved
    budgets = [n_steps, n_steps, 3, n_steps, 2, n_steps, 3, n_steps]           # by request: rows free up at different steps
    handles = {}
    for k, i in enumerate(order):
        handles[i] = eng.submit(reqs[i], budgets[i])
        if k >= 2:
            eng.step()                                              # one decode step between arrivals
    eng.run_until_idle()
    assert all(h.done.is_set() and h.error is None for h in handles.values()) and eng.failures == 0
    assert eng.joined_mid_flight >= 3
    n_engine = tokens_vs_fixture([handles[i].tokens for i in range(8)], g, "C4 engine", budgets=budgets)
    assert n_engine >= 10, n_engine

    from golden_inputs import vqa_request  # noqa: F401  (the share's images are rebuilt below with the same seeds)
    from PIL import Image
    imgs = [Image.fromarray(np.random.default_rng(s).integers(0, 256, (336, 336, 3), dtype=np.uint8)) for s in range(4)]
    http_proc = C4Processor(proc, reqs)
    httpd, backend = serve_continuous(ContinuousEngine(model, http_proc, slots=4, window=4096), port=0)
    threading.Thread(target=httpd.serve_forever, daemon=True).start()
    port, got = httpd.server_address[1], {}
    bodies = [{"prompt": f"c4:{i}", "max_tokens": n_steps, **({"images": [_png_uri(imgs[i])]} if i < 4 else {})} for i in range(8)]
    try:
        ths = [threading.Thread(target=lambda i=i, b=b: got.__setitem__(i, _post(port, b))) for i, b in enumerate(bodies)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    finally:
        httpd.shutdown()
        backend.close()
    toks = [[int(t) for t in got[i]["responses"][0].split()] for i in range(8)]
    n_http = tokens_vs_fixture(toks, g, "C4 HTTP")
    assert n_http >= 12, n_http
    print(f"C4 share: engine {n_engine}, HTTP {n_http} free-running tokens equal the per-request oracle's")
    del model, eng
    torch.cuda.empty_cache()


def test_engine_with_int8_kv_and_fp8_weights_vs_c5w_fixture():
    """BASELINE config 5 through the engine: `quantize_model=True` weights (fp8, weight-only
    arithmetic) and the int8 KV cache in the SLOT state; config 2's request, with a text request joining the other row between
    its decode steps, must produce the tokens of c5w_oracle.npz (an oracle applying the same quantisers) on every clear step."""
    from golden_inputs import vqa_request
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    g = np.load(GOLDEN + "/c5w_oracle.npz")
    model, proc = _full_model(g, quantized_fp8=True, use_quantized_cache=True, fp8_activations=False)
    inp = vqa_request(proc.img_processor, 0)
    inp["pixel_values"] = torch.from_numpy(inp["pixel_values"]).to("cuda:0")
    n_steps = g["tokens"].shape[1]
    eng = ContinuousEngine(model, proc, slots=2, window=4096)
    assert eng.st.quantized
    h = eng.submit(inp, n_steps)
    eng.step()                                                      # prefill + first decode step of the image request alone
    other = eng.submit({"input_ids": np.random.default_rng(9).integers(3, 32000, (1, 300)).astype(np.int64)}, 12)
    eng.run_until_idle()                                            # `other` is prefilled into the free row between h's steps
    assert h.error is None and other.error is None and eng.joined_mid_flight >= 1 and len(other.tokens) in range(1, 13)
    n = tokens_vs_fixture([h.tokens], g, "C5 engine")
    assert n >= 1
    print(f"C5 (weight-only fp8 + int8 KV) through the engine: {n} of {n_steps} tokens compared, all equal")
    del model, eng
    torch.cuda.empty_cache()


def test_long_rope_engine_vs_c3_fixture_and_regime_routing():
    """A request whose prompt + budget leaves the 4096-token window picks the LONG RoPE factors (phi.py:492): the router
    sends it to the long-window engine, whose rows use those factors -- tokens == c3_oracle.npz (5000-token prompt); a short
    request submitted at the same time runs on the short-factor engine (its tokens == a solo greedy run of the same model
    here: c1's fixture was generated under another lm_head)."""
    from phi_3_vision_mlx_amd.engine import ContinuousEngine, RegimeRouter
    g = np.load(GOLDEN + "/c3_oracle.npz")
    model, proc = _full_model(g, blind=True)
    ids = np.random.default_rng(4).integers(3, 32000, (1, 5000)).astype(np.int64)
    n_steps = g["tokens"].shape[1]
    router = RegimeRouter([ContinuousEngine(model, proc, slots=2, window=4096), ContinuousEngine(model, proc, slots=2, window=8192)])
    short_ids = np.random.default_rng(0).integers(3, 32000, (1, 128)).astype(np.int64)
    hs, hl = router.submit({"input_ids": short_ids}, n_steps), router.submit({"input_ids": ids}, n_steps)
    while router.safe_step() or router.waiting:
        pass
    assert hs.error is None and hl.error is None
    assert router.engines[0].steps > 0 and router.engines[1].steps > 0
    assert tokens_vs_fixture([hl.tokens], g, "long-RoPE engine") == n_steps          # c3: every step clear
    tok, cache = model.greedy_prefill(n_steps, input_ids=short_ids)
    solo = [int(tok.item())]
    for _ in range(n_steps - 1):
        _, tok = model.greedy_step(tok, cache)
        solo.append(int(tok.item()))
    assert hs.tokens[0] == solo[0]
    del model, router
    torch.cuda.empty_cache()


def test_mixed_batch_with_two_images_and_a_non_square_image_vs_the_oracle():
    """One decode batch of four requests no earlier test combined: a prompt with TWO images (both 640x480,
    different content; phi.py:400-415 advances `positions` per image), a 640x480 image alone (13 live crops), a 336x336 image and a text
    prompt -- length-bucketed prefill into one batch (dist.generate_requests), every row against the request's own B = 1 run of
    a LIVE oracle (itself pinned to the reference on exactly these image cases, tests/test_refmodel.py).  Tokens are compared on
    every clear step up to a row's first unclear one; every row's first token must be clear and equal."""
    import phi3v_oracle as orc
    from golden_inputs import make_image
    from phi_3_vision_mlx_amd import dist as pd
    from phi_3_vision_mlx_amd.api import load_synthetic
    spread, rel_tol, n = 4.0, 0.03, 5
    sq, land, land2 = make_image(336, 336, "noise", 0), make_image(640, 480, "smooth", 1), make_image(640, 480, "noise", 5)
    texts = ["<|user|>\n<|image_1|>\n<|image_2|>\nCompare the two.<|end|>\n<|assistant|>\n", "<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n",
             "<|user|>\n<|image_1|>\nAnd this one?<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
    imgs = [[land, land2], [land], [sq], None]       # two 640x480 images: 2 x 1921 image tokens, the batch stays in the short-RoPE regime
    for hs in range(40):                                      # a head under which every request's FIRST step is clear (host search, cheap)
        model, proc = load_synthetic(blind_model=False, tiny=True, seed=0, std_scale=4.0, device="cuda:0", lm_head_spread=spread, lm_head_seed=hs)
        oracle = orc.OraclePhi3V(model.cfg, {k: v.cpu() for k, v in model.w.items()}, cache_fp32=True)
        reqs = [proc(t, im) if im is not None else proc(t) for t, im in zip(texts, imgs)]
        assert max(int(np.asarray(r["input_ids"]).shape[-1]) for r in reqs) + n <= 4096      # ONE RoPE regime (phi.py:492), as per request
        norms = oracle.w["lm_head.weight"].float().norm(dim=-1).clamp_min(1e-30)
        ref_tok, ref_clear = [], []
        for r in reqs:
            tk, lg = orc.greedy_generate(oracle, dict(r), n, stop_on_eos=False)   # (the oracle takes the device crops to the host itself)
            lf = lg[0].float()
            v, i = lf.topk(2, dim=-1)
            E = rel_tol * (lf / norms).abs().amax(-1)
            ref_tok.append(tk[0].tolist()), ref_clear.append(((v[:, 0] - v[:, 1]) > E * (norms[i[:, 0]] + norms[i[:, 1]])).tolist())
        if all(c[0] for c in ref_clear):
            break
    else:
        pytest.fail("no head seed makes every first step clear")
    got = pd.generate_requests(model, proc, reqs, n, return_tokens=True)
    compared = 0
    for k, (toks, ref, clear) in enumerate(zip(got, ref_tok, ref_clear)):
        for step in range(n):
            if not clear[step]:
                break
            assert toks[step] == ref[step], f"request {k} step {step}: batch row {toks} != B = 1 oracle {ref}"
            compared += 1
    assert compared >= 8, compared
    print(f"mixed batch (two-image, 640x480, 336x336, text): {compared} tokens on clear steps equal the per-request oracle (head seed {hs})")


def _sharded_worker(rank, world, port, out_dir, backend="gloo"):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    local = rank if backend == "nccl" else 0                   # gloo: both ranks share the one GPU of the test box
    torch.cuda.set_device(local)
    if backend == "nccl":                                      # one rank per GPU over RCCL / xGMI (2-GPU boxes only)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_inputs import SERVE_PROMPTS, SERVE_STEPS, make_image
    from phi_3_vision_mlx_amd import dist as pd
    from phi_3_vision_mlx_amd.api import load_synthetic
    g = np.load(os.path.join(root, "tests", "golden", "tiny_serve_oracle.npz"))
    model, proc = load_synthetic(blind_model=False, tiny=True, seed=0, std_scale=4.0, device=f"cuda:{local}",
                                 lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]))
    images = [make_image(336, 336, "noise", 0)] + [None] * (len(SERVE_PROMPTS) - 1)
    mine = (SERVE_PROMPTS, images) if rank == 0 else (["junk"], None)
    got = pd.generate_sharded(*mine, preload=(model, proc), max_tokens=SERVE_STEPS, max_batch=2, return_tokens=True)
    if rank == 0:
        torch.save(got, os.path.join(out_dir, "sharded.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_generate_sharded_two_ranks_on_one_gpu_matches_the_oracle(tmp_path):
    """dist.generate_sharded on REAL kernels: two ranks (gloo rendezvous, both on the box's one GPU; on a node they would
    be one per GPU over RCCL) serve the 7 mixed requests (rank 0 holds the table; chunks of 2 rows; length-bucketed prefill);
    the request-ordered result == every request's B = 1 oracle run (tiny_serve_oracle.npz)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "sharded.pt")
    g = np.load(GOLDEN + "/tiny_serve_oracle.npz")
    assert len(got) == 7
    assert tokens_vs_fixture(got, g, "generate_sharded", min_first=2) >= 20


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI); the test box has one")


@needs_two_gpus
def test_generate_sharded_two_gpus_over_rccl_matches_the_oracle(tmp_path):
    """The same request table, one rank per GPU, backend "nccl" (= RCCL): the table is broadcast as device tensors, the
    results come back through RCCL collectives, and the request-ordered tokens == every request's B = 1 oracle run.  Skipped
    on 1-GPU boxes; on a multi-GPU node it runs with zero code changes."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    got = torch.load(tmp_path / "sharded.pt")
    g = np.load(GOLDEN + "/tiny_serve_oracle.npz")
    assert len(got) == 7
    assert tokens_vs_fixture(got, g, "generate_sharded over RCCL", min_first=2) >= 20


@needs_two_gpus
def test_bench_two_gpus_reports_its_rccl_ranks():
    """`bench.py --gpus 2` (the driver's scaling command, tiny model): one JSON line whose `rccl` object shows both ranks joined
    an all-reduce on DEVICE tensors over the nccl backend, with a per-rank rate each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--tiny", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["backend"] == "nccl" and line["rccl"]["ranks"] == 2
    assert len(line["rccl"]["per_rank_tokens_per_s"]) == 2 and min(line["rccl"]["per_rank_tokens_per_s"]) > 0


def test_bench_two_ranks_sharing_one_gpu():
    """The driver's multi-GPU command line (`python -m torch.distributed.run ... bench.py --gpus 2`) on a 1-GPU box: bench.py's debug
    switch P3V_BENCH_SHARE_GPU puts both ranks on GPU 0 over gloo (and plans every launch for a shared GPU).  One JSON line from rank 0,
    whole-job aggregate over both ranks, both ranks counted."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, P3V_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--tiny", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo"
    per_rank = line["rccl"]["per_rank_tokens_per_s"]
    assert len(per_rank) == 2 and min(per_rank) > 0
    assert line["value"] > 0 and line["steps"] == 8 and line["scaling"] == "weak"


def _fleet_worker(rank, world, port, out_dir):
    import os
    import sys
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.cuda.set_device(0)                                   # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_inputs import SERVE_STEPS, serve_requests
    from phi_3_vision_mlx_amd import fleet
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    g = np.load(os.path.join(root, "tests", "golden", "tiny_serve_oracle.npz"))
    model, proc = load_synthetic(blind_model=False, tiny=True, seed=0, std_scale=4.0, device="cuda:0",
                                 lm_head_spread=float(g["spread"][0]), lm_head_seed=int(g["head_seed"][0]))
    eng = ContinuousEngine(model, proc, slots=2, window=4096)
    groups = fleet.make_groups()
    if rank:
        fleet.worker(eng, groups)
        dist.destroy_process_group()
        return
    front = fleet.EngineFleet(eng, groups, world)
    stop = threading.Event()
    stepper = threading.Thread(target=front.serve_forever, args=(stop,), daemon=True)
    stepper.start()
    handles = [front.submit(r, SERVE_STEPS) for r in serve_requests(proc)]
    ok = all(h.done.wait(300) for h in handles)
    torch.save({"ok": ok, "errors": [repr(h.error) for h in handles], "tokens": [list(h.tokens) for h in handles],
                "sent": list(front.sent)}, os.path.join(out_dir, "fleet.pt"))
    front.close()
    stop.set()
    stepper.join(10)
    dist.destroy_process_group()


def test_engine_fleet_two_ranks_on_one_gpu_matches_the_oracle(tmp_path):
    """fleet.EngineFleet on REAL kernels: two ranks (gloo, both on the box's one GPU; on a node one per GPU), each stepping
    its own 2-slot continuous-batching engine; rank 0 dispatches the 7 mixed requests (the image request's pre-processed crops
    travel to the other rank when it is picked); every request's tokens == its own B = 1 oracle run, whichever rank ran it."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_fleet_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "fleet.pt")
    assert got["ok"] and all(e == "None" for e in got["errors"]), got["errors"]
    assert sum(got["sent"]) == 7 and min(got["sent"]) >= 2, got["sent"]     # both engines served requests
    g = np.load(GOLDEN + "/tiny_serve_oracle.npz")
    assert tokens_vs_fixture(got["tokens"], g, "engine fleet", min_first=2) >= 20


def test_server_cli_under_torchrun_two_ranks_serves_http(tmp_path):
    """The command INTEGRATION.md gives for a node -- `python -m torch.distributed.run --nproc-per-node 2 -m ...server
    --continuous` -- on the tiny synthetic model (both ranks on the box's one GPU): rank 0 answers HTTP, rank 1 runs
    fleet.worker; eight concurrent text requests all come back 200 with text, and the same prompt gives the same text
    whichever rank served it."""
    import json
    import os
    import signal
    import socket
    import subprocess
    import sys
    import threading
    import time
    import urllib.request
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p
    http_port, master_port = free_port(), free_port()
    log = open(tmp_path / "server.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                             "--master-port", str(master_port), "-m", "phi_3_vision_mlx_amd.server", "--continuous", "--synthetic", "--tiny",
                             "--port", str(http_port), "--slots", "2"], cwd=root, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        def post(prompt, max_tokens=6):
            req = urllib.request.Request(f"http://127.0.0.1:{http_port}/v1/completions", method="POST",
                                         data=json.dumps({"prompt": prompt, "max_tokens": max_tokens}).encode(),
                                         headers={"Content-Type": "application/json"})
            with urllib.request.urlopen(req, timeout=120) as r:
                return r.status, json.loads(r.read())
        deadline = time.time() + 240
        while True:                                            # wait for the listener (both ranks load the model first)
            assert proc.poll() is None, open(tmp_path / "server.log").read()[-2000:]
            try:
                socket.create_connection(("127.0.0.1", http_port), timeout=1).close()
                break
            except OSError:
                assert time.time() < deadline, open(tmp_path / "server.log").read()[-2000:]
                time.sleep(0.5)
        prompts = [f"question number {i % 4} about the weather" for i in range(8)]      # every prompt twice
        out = [None] * len(prompts)

        def one(i):
            out[i] = post(prompts[i])
        threads = [threading.Thread(target=one, args=(i,)) for i in range(len(prompts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(180)
        assert all(o is not None and o[0] == 200 for o in out), out
        texts = [o[1]["responses"][0] for o in out]                                     # the reference's response shape (server.py:22-25)
        assert all(isinstance(t, str) for t in texts) and all(texts[i] == texts[i + 4] for i in range(4)), texts
    finally:
        os.killpg(proc.pid, signal.SIGTERM)                    # the process group this test started (torchrun + its two ranks)
        try:
            proc.wait(30)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
        log.close()
