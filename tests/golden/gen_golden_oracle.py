"""Golden vectors from the CPU oracle (oracle/phi3v_oracle.py) on seeded synthetic weights.

Every fixture stores, per greedy step, the oracle's token, the FULL last-position logits (bf16 bit patterns) and the
top-2 margin as a fraction of max|logit|.  The lm_head is the decisive-argmax head of `weights.peaked_lm_head`
(random rows x seeded power-of-two row scales); its seed is SEARCHED so that every step of the fixture has a margin
> 4 x REL_TOL -- greedy token ids can then be asserted exactly on every step (n_clear == n_steps), which a plain
N(0, 0.02) head never allows (its top-2 gap is a few bf16 ulps of the logits; tools/precision_study.py).

  tiny_oracle.npz   tiny text / batch / vision models + choose / constrain traces (decision margins recorded)
  tiny_serve_oracle.npz  7 mixed requests on the tiny vision model, each at B = 1 under one head (serving / sharding tests)
  c1_oracle.npz     FULL-SIZE Phi-3-mini-128K, BASELINE config 1 (128-token prompt, text-only): prefill + 5 decode steps
  c{1,2}_long_oracle.npz  the same requests over the benchmark's horizon (128 / 32 greedy steps), under the decisive head AND
                    the plain N(0, 0.02) head, compact per-step records (`long_fixture`)
  c2_oracle.npz     FULL-SIZE Phi-3-Vision, BASELINE config 2 = bench.py's rank-0 request (2531-token prompt): prefill + 3 steps
  c4_oracle.npz     FULL-SIZE, one GPU's share of BASELINE config 4 (4 image + 4 text requests), each run on its own at
                    B = 1 (the reference's only image path, phi_3_vision_mlx.py:377-378): prefill + 3 steps per request
  c3_oracle.npz     FULL-SIZE text model, 5000-token prompt (LONG RoPE factors, BASELINE config 3's path): prefill + 3 steps
  c5_oracle.npz     config 2's request on the oracle with config 5's quantisers applied (e4m3 weights with per-row scales,
                    e4m3 activations with per-token scales in the prompt-sized projections, int8 KV with per-token scales;
                    see `c5`)

    python tests/golden/gen_golden_oracle.py [tiny|full|c5|all]        (full = c1 + c2 + c4 in one process, ~40 GB RAM)
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
from golden_inputs import c4_share, make_image, vqa_request  # noqa: E402
from phi_3_vision_mlx_amd.config import make_config, phi3v_config_dict, tiny_config_dict  # noqa: E402
from phi_3_vision_mlx_amd.processor import Phi3FProcessor, Phi3VProcessor  # noqa: E402
from phi_3_vision_mlx_amd.weights import peaked_lm_head, synth_weights  # noqa: E402

# TOLERANCE MODEL.  logit_v = W_v . h, so an error dh of the final hidden state moves entry v by at most |W_v| |dh|: the
# natural unit of a logit error is the entry's own lm_head ROW NORM n_v.  With z_v = logit_v / n_v the GPU tests assert, on
# EVERY vocabulary entry,      |HIP_v - oracle_v| <= rel_tol * max_u |z_u| * n_v   (+ one bf16 ulp of the entry),
# i.e. the classic "within rel_tol of max|logit|" in the row-normalised space (for a plain N(0, s) head, whose rows all have
# the same norm, the two are the same statement).  A step is CLEAR when the oracle's top-2 margin exceeds the SUM of the two
# entries' tolerances -- then any implementation that meets the tolerance must pick the same token; statistically that
# margin is > 6 sigma of the difference of the two errors (the tolerance bounds the worst of 32064 entries, ~4.1 sigma).
# rel_tol is what two CORRECT implementations of this bf16 model differ by (measured HIP vs oracle, z-space, worst entry):
#   2 layers (tiny) < 2 %; 32 layers: 2.7-3.1 % on 2531-token image prompts, 6.0-7.4 % on 65-233-token text prompts (few keys
#   -> little averaging in the attention); the oracle against ITSELF with float64 accumulation already differs by 2.4 % after
#   24 layers (tools/precision_study.py): the bf16 residual stream, not the build's bf16 attention operands, sets the floor.
REL_TOL = 0.03            # tiny (2-layer) fixtures
REL_TOL_LONG = 0.045      # full size, 2531-token image prompts
REL_TOL_SHORT = 0.09      # full size, short text prompts
CKPT = os.environ.get("P3V_ORACLE_CKPT", "/tmp/p3v_oracle_ckpt_r4")     # prefilled requests are kept here between runs (GBs)
SPREAD = 4.0                   # log2-sd of the lm_head row scales
C1_STEPS = 6                   # prefill + 5 decode steps, all clear (each extra all-clear step costs ~4x more head seeds)
REL_TOL_WC = 0.012             # full size, WELL-CONDITIONED checkpoint (residual branches x 1 / 1024; measured HIP-vs-reference: 0.93 %): gen_golden_refmodel.py wc
REL_TOL_C3 = 0.08              # full size, 5000-token text prompt (long RoPE factors): 7 % until RMSNorm rounded twice (round 4; measured 7.5 %)
REL_TOL_C5 = 0.25              # config 5, W8A8 prefill + W8A16 decode + int8 KV: see c5()
REL_TOL_C5W = 0.07             # config 5 with fp8_activations=False (weight-only fp8 + int8 KV)
BF16, F32 = torch.bfloat16, torch.float32


def bits(lg):
    return lg.to(BF16).contiguous().view(torch.int16).numpy().view(np.uint16)


def row_norms(head):
    """fp32 L2 norms of the lm_head rows (of the values the projection really multiplies by)."""
    return head.to(F32).norm(dim=-1).clamp_min(1e-30)


def clearance(lg, n, rel_tol):
    """(top-2 margin) / (sum of the two entries' tolerances) per row; > 1 = the step is clear.  rel_tol: float or [B]."""
    lf = lg.to(F32)
    v, i = lf.topk(2, dim=-1)
    E = torch.as_tensor(rel_tol, dtype=F32) * (lf / n).abs().amax(dim=-1)
    return (v[..., 0] - v[..., 1]) / (E * (n[i[..., 0]] + n[i[..., 1]]))


class Prefilled:
    """A request after the oracle's prefill: last-position hidden state + KV cache; decode steps can be replayed for any
    lm_head by rewinding `offset` (KVCache semantics, phi.py:589-591)."""

    def __init__(self, o, inputs, n_steps, tag=None, hidden_file=None):
        """hidden_file: also dump the residual stream after every layer (last 32 positions, bf16) for
        tools/precision_decomp.py's HIP half (measured per-layer |HIP - oracle|; git-ignored, travels to the GPU box)."""
        self.o, self.inputs = o, inputs
        t0 = time.time()
        f = os.path.join(CKPT, f"{tag}_n{n_steps}.pt") if tag else None
        if f and os.path.exists(f):
            d = torch.load(f)
            self.h0, self.S, self.masker, self.roper = d["h0"], d["S"], orc.OracleMask4D(1, None), d["roper"]
            self.masker.allowed = d["allowed"]
            self.cache = []
            for kv in d["kv"]:
                c = orc.OracleKVCache(o.cfg, kv.shape[1], self.S, n_steps)
                c.kv, c.offset = kv, self.S
                self.cache.append(c)
            print(f"  prefill S={self.S} loaded from {f}", flush=True)
            return
        hs = []
        x, self.cache = o.backbone(inputs["input_ids"], inputs.get("pixel_values"), inputs.get("image_sizes"),
                                   inputs.get("positions"), None, inputs.get("pids"), inputs.get("mask"), n_steps, None, 1,
                                   hidden_hook=(lambda i, h: hs.append(h[0, -32:].clone())) if hidden_file else None)
        if hidden_file:
            os.makedirs(os.path.dirname(hidden_file), exist_ok=True)
            np.savez(hidden_file, hidden_bf16=torch.stack(hs).to(BF16).view(torch.int16).numpy(),
                     h0_bf16=x[0, -1].to(BF16).view(torch.int16).numpy())
        self.h0 = x[:, -1:, :].clone()
        self.S = self.cache[0].offset
        self.masker, self.roper = o._masker, o._roper
        print(f"  prefill S={self.S} B={x.shape[0]}: {time.time() - t0:.0f}s", flush=True)
        if f:
            os.makedirs(CKPT, exist_ok=True)
            torch.save({"h0": self.h0, "S": self.S, "allowed": self.masker.allowed, "roper": self.roper, "kv": [c.kv for c in self.cache]}, f)

    rel_tol = REL_TOL

    def greedy(self, head_f32, n_steps, need_clear_steps=None, teacher=None, norms=None):
        """Greedy steps under lm_head `head_f32`; stops early (returns None) when one of the first `need_clear_steps`
        steps is not clear."""
        norms = row_norms(head_f32) if norms is None else norms
        o = self.o
        o._masker, o._roper = self.masker, self.roper
        for c in self.cache:
            c.offset = self.S
        h, toks, lgs, mgs = self.h0, [], [], []
        for t in range(n_steps):
            lg = orc._linear(h, head_f32)[:, -1]
            m = clearance(lg, norms, self.rel_tol)
            if need_clear_steps is not None and t < need_clear_steps and m.min().item() <= 1.0:
                return None
            tok = torch.argmax(lg.to(F32), dim=-1)[:, None]
            toks.append(tok), lgs.append(lg), mgs.append(m)
            if t + 1 < n_steps:
                h, _ = o.backbone(tok if teacher is None else teacher[:, t:t + 1], None, None, None, self.cache,
                                  self.inputs.get("pids"), self.inputs.get("mask"), 0, None, 1)
        return torch.cat(toks, 1), torch.stack(lgs, 1), torch.stack(mgs, 1)


def conditioning(r, head_f32, toks, n_steps):
    """How far every decode step of a fixture moves when the rotated q / k are rounded to bf16 (teacher-forced on `toks`):
    per step, max_v |z_v - z'_v| / max_v |z_v| -- the fixture tolerance's own unit."""
    norms = row_norms(head_f32)
    base = r.greedy(head_f32, n_steps, teacher=toks, norms=norms)[1].to(F32)
    plain_rot = orc.rotate_half
    orc.rotate_half = lambda x, c, s_: plain_rot(x, c, s_).to(BF16).to(F32)
    try:
        pert = r.greedy(head_f32, n_steps, teacher=toks, norms=norms)[1].to(F32)
    finally:
        orc.rotate_half = plain_rot
    return [float((((base[:, t] - pert[:, t]).abs() / norms).amax(-1) / (base[:, t] / norms).abs().amax(-1)).max()) for t in range(n_steps)]


def search_head(reqs, base_head, n_steps, max_seeds=20000, first_seed=0, need="all", min_distinct=1):
    """Smallest lm_head seed for which the requests' greedy runs are clear.  need = "all": every step of every request;
    need = "prefill": the first step of every request (then the decode steps are taken as they come); an int: that many steps.
    min_distinct: a greedy run that repeats one token is a weak witness -- ask for some variety."""
    base = base_head.to(F32)
    for hs in range(first_seed, first_seed + max_seeds):
        head = peaked_lm_head(base, SPREAD, hs)
        norms = row_norms(head)
        if need != 0 and min(clearance(orc._linear(r.h0, head)[:, -1], norms, r.rel_tol).min().item() for r in reqs) <= 1.0:
            continue                                        # cheap filter: the prefill step of every request
        out = []
        for r in reqs:
            res = r.greedy(head, n_steps, need_clear_steps=n_steps if need == "all" else 1 if need == "prefill" else int(need), norms=norms)
            if res is None:
                break
            out.append(res)
        if len(out) == len(reqs) and min(len(set(r[0].reshape(-1).tolist())) for r in out) >= min_distinct:
            return hs, out
    raise RuntimeError("no lm_head seed with clear margins found")


def pack(prefix, hs, res, out):
    toks, lgs, mgs = res
    out[prefix + "head_seed"] = np.asarray([hs], dtype=np.int32)
    out[prefix + "tokens"] = toks.numpy().astype(np.int32)
    out[prefix + "logits_bf16"] = bits(lgs)
    out[prefix + "margins"] = mgs.numpy().astype(np.float32)
    print(f"  {prefix}: head_seed {hs}, tokens {toks.tolist()}, min clearance {mgs.min().item():.3f}", flush=True)


N_SAMPLE = 256                 # vocabulary entries kept per step by the long-horizon fixtures (+ the top 8)
C1_LONG, C2_LONG = 128, 32     # BASELINE config 1's 128 generated tokens; the reference's benchmark() uses 100 (:1272)


def pack_long(prefix, head, res, out, rel_tol):
    """Compact per-step record for long greedy runs (a full logits row per step would be 64 KB): the token, the top-8 entries,
    N_SAMPLE seeded vocabulary entries, max_u |z_u| (the tolerance unit needs it), the fp32 log-sum-exp and the clearance."""
    toks, lgs, mgs = res
    lf = lgs.to(F32)
    n = row_norms(head)
    sample = torch.from_numpy(np.sort(np.random.default_rng(1234).choice(lf.shape[-1], N_SAMPLE, replace=False)))
    v, i = torch.sort(lf, dim=-1, descending=True, stable=True)
    out[prefix + "tokens"] = toks.numpy().astype(np.int32)
    out[prefix + "top_ids"] = i[..., :8].numpy().astype(np.int32)
    out[prefix + "top_bf16"] = bits(v[..., :8])
    out[prefix + "sample_ids"] = sample.numpy().astype(np.int32)
    out[prefix + "sample_bf16"] = bits(lf[..., sample])
    out[prefix + "zmax"] = (lf / n).abs().amax(-1).numpy().astype(np.float32)
    out[prefix + "lse"] = torch.logsumexp(lf, dim=-1).numpy().astype(np.float32)
    out[prefix + "margins"] = mgs.numpy().astype(np.float32)
    clear = (mgs > 1.0)
    print(f"  {prefix or 'long'}: {toks.shape[1]} steps, {int(clear.sum())} clear, {len(set(toks.reshape(-1).tolist()))} distinct tokens, "
          f"rel_tol {rel_tol}", flush=True)


def long_fixture(name, r, base, hs, n_steps, rel_tol):
    """`<name>_long_oracle.npz`: the oracle's own greedy run over the benchmark's horizon under (a) the fixture's decisive head
    (seed hs: its first steps are the short fixture's) and (b) the PLAIN N(0, 0.02) head bench.py times -- no seed search, no
    peaking: steps are clear or not as they come, the GPU test reports 'exact on k of n clear steps' and bounds every step's
    logits."""
    out = dict(COMMON, rel_tol=np.asarray([rel_tol], dtype=np.float32), head_seed=np.asarray([hs], dtype=np.int32),
               n_ids=np.asarray([r.S], dtype=np.int32))
    r.rel_tol = rel_tol
    for prefix, head in (("peaked_", peaked_lm_head(base.to(F32), SPREAD, hs)), ("plain_", base.to(F32))):
        t0 = time.time()
        res = r.greedy(head, n_steps)
        pack_long(prefix, head, res, out, rel_tol)
        print(f"    {time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, f"{name}_long_oracle.npz"), **out)
    print("wrote", f"{name}_long_oracle.npz", flush=True)


COMMON = dict(rel_tol=np.asarray([REL_TOL], dtype=np.float32), spread=np.asarray([SPREAD], dtype=np.float32))
COMMON_LONG = dict(COMMON, rel_tol=np.asarray([REL_TOL_LONG], dtype=np.float32))
COMMON_SHORT = dict(COMMON, rel_tol=np.asarray([REL_TOL_SHORT], dtype=np.float32))
TINY_PROMPTS = ["<|user|>\nPick A or B.<|end|>\n<|assistant|>\n", "<|user|>\nName a colour of the sky.<|end|>\n<|assistant|>\n"]
TINY_VIS_PROMPT = "<|user|>\n<|image_1|>\nWhat is shown?<|end|>\n<|assistant|>\n"
CONSTRAINT = (3, " The answer is")


def tiny():
    out = dict(COMMON)
    for blind in (True, False):
        cfg = make_config(tiny_config_dict(vision=not blind))
        w = synth_weights(cfg, seed=0, std_scale=4.0)
        base = w["lm_head.weight"]
        o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
        proc = (Phi3FProcessor if blind else Phi3VProcessor)(None)

        def with_head(hs):
            o.w["lm_head.weight"] = peaked_lm_head(base, SPREAD, hs)
            o._f32.pop("lm_head.weight", None)
        if blind:
            ids = np.random.default_rng(11).integers(3, 32000, (1, 40)).astype(np.int64)
            out["text_ids"] = ids
            hs, (res,) = search_head([Prefilled(o, {"input_ids": ids}, 8)], base, 8, min_distinct=3)
            pack("text_", hs, res, out)
            inp = proc(TINY_PROMPTS)
            hs, (res,) = search_head([Prefilled(o, inp, 6)], base, 6, min_distinct=3)
            pack("batch_", hs, res, out)
            # choose + constrain: seeds searched so that no decision of the loops hinges on a near-tie (trace margins)
            opts = proc([f" {c}" for c in "ABCDE"])["input_ids"][:, -1]
            idc = proc.tokenizer.encode(CONSTRAINT[1], add_special_tokens=False)[1:]
            for hs in range(4000):
                with_head(hs)
                lg, _ = o(**proc(TINY_PROMPTS), max_tokens=0)
                oi = torch.as_tensor(opts).long()
                lp, nn = lg[:, -1].to(F32)[:, oi], row_norms(o.w["lm_head.weight"])
                v2, i2 = lp.topk(2, dim=-1)
                E = REL_TOL * (lg[:, -1].to(F32) / nn).abs().amax(-1)
                if ((v2[:, 0] - v2[:, 1]) / (E * (nn[oi][i2[:, 0]] + nn[oi][i2[:, 1]]))).min().item() > 1.0:   # option gap > their tolerances
                    out["choose_head_seed"] = np.asarray([hs], dtype=np.int32)
                    out["choose_idx"] = np.asarray(orc.choose_from(o, proc(TINY_PROMPTS), opts), dtype=np.int32)
                    print("  choose: head_seed", hs, out["choose_idx"].tolist(), flush=True)
                    break
            else:
                raise RuntimeError("no clear choose seed")
            with_head(int(out["choose_head_seed"][0]))                    # regression vectors of the constrain loop (the GPU test
            for ub in (False, True):                                      # walks the decisions of a live oracle run instead)
                cin = dict(inp) if not ub else proc(TINY_PROMPTS[:1] * 2)     # beams: two identical rows (per-row pids needed)
                s, sc = orc.constrain_one(o, dict(cin), CONSTRAINT, idc, use_beam=ub)
                out[f"constrain_beam{int(ub)}_synth"] = s.numpy().astype(np.int32)
                out[f"constrain_beam{int(ub)}_score"] = sc.float().numpy()
        else:
            inp = proc(TINY_VIS_PROMPT, [make_image(336, 336, "noise", 0)])
            out["vis_n_ids"] = np.asarray([np.asarray(inp["input_ids"]).shape[1]], dtype=np.int32)
            hs, (res,) = search_head([Prefilled(o, inp, 4)], base, 4, min_distinct=2)
            pack("vis_", hs, res, out)
    np.savez_compressed(os.path.join(HERE, "tiny_oracle.npz"), **out)
    print("wrote tiny_oracle.npz")


def tiny_serve():
    """tiny_serve_oracle.npz: the mixed request set of golden_inputs.SERVE_TEXTS (1 image + 6 text prompts, 4..2500+ tokens)
    on the tiny VISION model, every request run on its own at B = 1 (the reference's only image path,
    phi_3_vision_mlx.py:377-378) under ONE lm_head: SERVE_STEPS greedy tokens + clearance per (request, step).  The head
    seed makes the first 2 steps of every request clear (no seed in 20000 clears 3); later steps are taken as they come (a test compares a request's tokens up to its
    first unclear step).  Pins the continuous-batching engine, the HTTP handler on it, dist.prefill_requests and
    dist.generate_sharded to the oracle."""
    from golden_inputs import SERVE_STEPS, serve_requests
    cfg = make_config(tiny_config_dict(vision=True))
    w = synth_weights(cfg, seed=0, std_scale=4.0)
    base = w["lm_head.weight"]
    o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    reqs = [Prefilled(o, {k: (torch.from_numpy(np.asarray(v)) if k == "pixel_values" else v) for k, v in r.items()}, SERVE_STEPS)
            for r in serve_requests(Phi3VProcessor(None))]
    hs, results = search_head(reqs, base, SERVE_STEPS, need=2, min_distinct=2)
    out = dict(COMMON, head_seed=np.asarray([hs], dtype=np.int32), n_ids=np.asarray([r.S for r in reqs], dtype=np.int32))
    out["tokens"] = np.concatenate([r[0].numpy() for r in results]).astype(np.int32)              # [7, SERVE_STEPS]
    out["margins"] = np.concatenate([r[2].numpy() for r in results]).astype(np.float32)
    print(f"  tiny serve: head_seed {hs}, clear {(out['margins'] > 1.0).sum()} of {out['margins'].size}, tokens {out['tokens'].tolist()}", flush=True)
    np.savez_compressed(os.path.join(HERE, "tiny_serve_oracle.npz"), **out)
    print("wrote tiny_serve_oracle.npz")


def _full_oracle(transform=None, outliers=None, residual_scale=None):
    torch.set_num_threads(8)
    cfg = make_config(phi3v_config_dict(vision=True))
    t0 = time.time()
    w = synth_weights(cfg, seed=0, outliers=outliers, residual_scale=residual_scale)
    print(f"weights {time.time() - t0:.0f}s", flush=True)
    base = w["lm_head.weight"]
    if transform is not None:
        w = transform(cfg, w)
    return cfg, orc.OraclePhi3V(cfg, w, cache_fp32=True), base


def full():
    """c1 + c2 + c4 on ONE oracle: the blind model's decoder weights are the vision model's (hash-seeded by tensor name)."""
    cfg, o, base = _full_oracle()
    ip = Phi3VProcessor(None).img_processor
    ids = np.random.default_rng(0).integers(3, 32000, (1, 128)).astype(np.int64)
    print("c1", flush=True)
    r1 = Prefilled(o, {"input_ids": ids}, C1_LONG, tag="c1")
    r1.rel_tol = REL_TOL_SHORT
    hs, (res,) = search_head([r1], base, C1_STEPS, min_distinct=2)
    out = dict(COMMON_SHORT, ids=ids)
    pack("", hs, res, out)
    np.savez_compressed(os.path.join(HERE, "c1_oracle.npz"), **out)
    long_fixture("c1", r1, base, hs, C1_LONG, REL_TOL_SHORT)
    del r1
    print("c4 share (request 0 = c2)", flush=True)
    share = c4_share(ip)
    reqs = [Prefilled(o, {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in r.items()}, C2_LONG if i == 0 else 4, tag=f"c4r{i}")
            for i, r in enumerate(share)]
    for r, rq in zip(reqs, share):
        r.rel_tol = REL_TOL_LONG if "pixel_values" in rq else REL_TOL_SHORT
    hs, (res,) = search_head(reqs[:1], base, 4, min_distinct=2)
    out = dict(COMMON_LONG, n_ids=np.asarray([reqs[0].S], dtype=np.int32))
    pack("", hs, res, out)
    np.savez_compressed(os.path.join(HERE, "c2_oracle.npz"), **out)
    long_fixture("c2", reqs[0], base, hs, C2_LONG, REL_TOL_LONG)
    for r, rq in zip(reqs, share):
        r.rel_tol = REL_TOL_LONG if "pixel_values" in rq else REL_TOL_SHORT
    hs, results = search_head(reqs, base, 4, need="prefill")
    out = dict(COMMON, rel_tol=np.asarray([r.rel_tol for r in reqs], dtype=np.float32),
               n_ids=np.asarray([r.S for r in reqs], dtype=np.int32), head_seed=np.asarray([hs], dtype=np.int32))
    out["tokens"] = np.concatenate([r[0].numpy() for r in results]).astype(np.int32)              # [8, 4]
    out["logits_bf16"] = np.concatenate([bits(r[1]) for r in results])                            # [8, 4, V]
    out["margins"] = np.concatenate([r[2].numpy() for r in results]).astype(np.float32)
    print(f"  c4: head_seed {hs}, clear {(out['margins'] > 1.0).sum()} of {out['margins'].size}", flush=True)
    np.savez_compressed(os.path.join(HERE, "c4_oracle.npz"), **out)
    print("wrote c1 / c2 / c4")


def c3():
    """BASELINE config 3's distinguishing path at a size the O(S^2) oracle can hold: a 5000-token text prompt, so that
    S + max_tokens > 4096 selects the LONG RoPE factors (phi.py:492) -- full-size model, prefill + 3 decode steps, all clear.
    (32768 tokens would need 137 GB of fp32 scores per layer in the reference's formulation; the GPU tests cover that
    size through properties.)"""
    cfg, o, base = _full_oracle()
    ids = np.random.default_rng(4).integers(3, 32000, (1, 5000)).astype(np.int64)
    r = Prefilled(o, {"input_ids": ids}, 4, tag="c3")
    r.rel_tol = REL_TOL_C3                                     # random TEXT ids: measured 5.3 % (image prompts of half the length: 3 %)
    hs, (res,) = search_head([r], base, 4, min_distinct=2)
    out = dict(COMMON, rel_tol=np.asarray([REL_TOL_C3], dtype=np.float32), n_ids=np.asarray([r.S], dtype=np.int32))
    pack("", hs, res, out)
    np.savez_compressed(os.path.join(HERE, "c3_oracle.npz"), **out)
    print("wrote c3_oracle.npz")


def c5_quantisers(cfg, w):
    """Config 5's weight quantiser applied to the oracle's weights: decoder projections + lm_head -> e4m3 with one fp32
    scale per output row (ops.quantize_fp8_rows), kept as the exact fp32 products."""
    from phi_3_vision_mlx_amd.ops import quantize_fp8_rows
    out = dict(w)
    for k in w:
        if k.startswith("model.layers.") and k.endswith("_proj.weight"):
            w8, sc = quantize_fp8_rows(w[k])
            out[k] = w8.view(torch.float8_e4m3fn).to(F32) * sc[:, None]
    return out


def quantize_act_rows(x):
    """Config 5's activation quantiser (csrc k_quant_fp8_rows) on what the build feeds it -- the bf16 value of the
    projection's input: one scale s per token row = max|x| / 448, codes e4m3(x * (1 / s)); returns the dequantised rows."""
    xb = x.to(BF16).to(F32)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (xb * (1.0 / s)).to(torch.float8_e4m3fn).to(F32) * s


def c5_proj(o, act8=True):
    """nn.Linear of the config-5 model: e4m3 x scale weights (already folded into o.W); prompt-sized inputs (more than 16
    rows: the build's MFMA path) get their activations quantised too (W8A8), decode-sized ones stay bf16 (W8A16 GEMV)."""
    def proj(x, name):
        W = o.W(name)
        xe = quantize_act_rows(x) if act8 and x.shape[0] * x.shape[1] > 16 else x.to(F32)
        y = xe @ W.to(F32).t()
        return y.to(BF16) if x.dtype == BF16 else y
    return proj


def quantize_kv_rows(x):
    """Config 5's KV quantiser (csrc k_kv_quantize): per (row, head, token) scale = max|x| / 127, code = rne(x / scale)
    clamped to [-127, 127]; returns the dequantised values."""
    amax = x.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    return torch.clamp(torch.round(x / scale), -127, 127) * scale


class QuantKVCache(orc.OracleKVCache):
    """KVCache with config 5's int8 storage, as the build keeps it: a PROMPT-sized call (more than 16 new tokens) attends to
    its exact keys / values and stores their quantised copy (the prompt stays exact during the prefill, phi.py:531-533);
    a decode-sized call quantises its new rows FIRST and attends over what it stores (csrc k_attn_decode_q8*: one code
    path in the tile, phi.py:545-546)."""

    def __call__(self, keys, values, n_beam):
        start = self.offset
        if keys.shape[2] <= 16:
            return super().__call__(quantize_kv_rows(keys.to(BF16).to(F32)), quantize_kv_rows(values.to(BF16).to(F32)), n_beam)
        k, v = super().__call__(keys, values, n_beam)
        k, v = k.clone(), v.clone()                                    # this call attends to the exact new rows ...
        self.kv[0, :, :, start:self.offset] = quantize_kv_rows(self.kv[0, :, :, start:self.offset].to(BF16).to(F32))
        self.kv[1, :, :, start:self.offset] = quantize_kv_rows(self.kv[1, :, :, start:self.offset].to(BF16).to(F32))
        return k, v                                                    # ... later calls read the quantised ones


def c5(act8=True):
    """act8=True: the default config-5 path (W8A8 prompt projections on the fp8 MFMA) -> c5_oracle.npz; False: weight-only
    fp8 (`fp8_activations=False`: dequantise + bf16 MFMA) -> c5w_oracle.npz.  Both with the int8 KV cache.
    Why the W8A8 tolerance is so wide: an e4m3 code is a 6-12 % step, so wherever two correct implementations feed a
    quantiser values that differ by eps (the build's bf16 attention output vs the oracle's fp32 one: 0.4 %), eps / 9 % of
    the codes flip by a whole step -- rounding noise sqrt(eps x 9 %) instead of eps, at every quantiser of every layer.
    Measured: 2 layers 4-9 % (tests: test_c5_quantisers_small_model_tight), 32 layers 18 % of max|z|."""
    cfg, o, base = _full_oracle(c5_quantisers)
    o.proj = c5_proj(o, act8)
    from phi_3_vision_mlx_amd.ops import quantize_fp8_rows
    orig = orc.OracleKVCache
    orc.OracleKVCache = QuantKVCache
    try:
        ip = Phi3VProcessor(None).img_processor
        inp = vqa_request(ip, 0)
        r = Prefilled(o, {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}, 4, tag="c5" if act8 else "c5w")

        global peaked_lm_head
        plain = peaked_lm_head

        def q_head(b, spread, hs):                                    # the peaked head goes through the weight quantiser too
            w8, sc = quantize_fp8_rows(plain(b.to(BF16), spread, hs))
            return w8.view(torch.float8_e4m3fn).to(F32) * sc[:, None]
        r.rel_tol = REL_TOL_C5 if act8 else REL_TOL_C5W
        peaked_lm_head = q_head
        try:
            # W8A8 noise is large (REL_TOL_C5): an all-clear run is out of reach -- the first step must be clear, the decode
            # steps are taken as they come and the test asserts the tokens of the clear ones
            hs, (res,) = search_head([r], base, 4, need="prefill")
        finally:
            peaked_lm_head = plain
    finally:
        orc.OracleKVCache = orig
    out = dict(COMMON, rel_tol=np.asarray([r.rel_tol], dtype=np.float32), n_ids=np.asarray([r.S], dtype=np.int32))
    pack("", hs, res, out)
    name = "c5_oracle.npz" if act8 else "c5w_oracle.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


REL_TOL_C5_WC = 0.03            # VERDICT r05 item 1c asked for <= 8 % (config 5 W8A8 on the well-conditioned checkpoint); measured HIP-vs-oracle 1.6 %
REL_TOL_C5W_WC = 0.02           # ... and weight-only fp8 (W8A16): measured 1.0 %


def c5_wc(act8=True):
    """`c5_wc_oracle.npz` / `c5w_wc_oracle.npz` (round 6): config 5's three quantisers (e4m3 weights, e4m3 prompt activations when
    act8, int8 KV) applied to the oracle on the WELL-CONDITIONED checkpoint (decoder residual branches x 1 / 1024, as
    gen_golden_refmodel.wc) for BASELINE config 2's request, 8 greedy steps under two UNSEARCHED heads (plain, peaked seed 0; the
    peaked one through the weight quantiser, as the build quantises its lm_head).  The reference has no fp8 path: this fixture is
    the oracle's, like every c5 fixture.  On this net a flipped e4m3 code is not amplified by the 32 layers behind it, so what two
    correct implementations of the SAME quantised arithmetic differ by is visible at its own size (c5_oracle.npz on the plain
    checkpoint needs 25 %)."""
    from phi_3_vision_mlx_amd.ops import quantize_fp8_rows
    n_steps = 8
    rel_tol = REL_TOL_C5_WC if act8 else REL_TOL_C5W_WC
    cfg, o, base = _full_oracle(c5_quantisers, residual_scale=1.0 / 1024)
    o.proj = c5_proj(o, act8)
    orig = orc.OracleKVCache
    orc.OracleKVCache = QuantKVCache
    try:
        ip = Phi3VProcessor(None).img_processor
        inp = vqa_request(ip, 0)
        r = Prefilled(o, {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}, n_steps, tag="c5wc" if act8 else "c5wwc")
        r.rel_tol = rel_tol

        def q(h):                                                    # the build quantises its lm_head rows too
            w8, sc = quantize_fp8_rows(h.to(BF16))
            return w8.view(torch.float8_e4m3fn).to(F32) * sc[:, None]
        out = dict(COMMON, rel_tol=np.asarray([rel_tol], dtype=np.float32), n_ids=np.asarray([r.S], dtype=np.int32),
                   residual_scale=np.asarray([1.0 / 1024], dtype=np.float32))
        for prefix, head in (("plain_", q(base)), ("peaked0_", q(peaked_lm_head(base.to(F32), SPREAD, 0)))):
            res = r.greedy(head, n_steps)
            pack_long(prefix, head, res, out, rel_tol)
    finally:
        orc.OracleKVCache = orig
    name = "c5_wc_oracle.npz" if act8 else "c5w_wc_oracle.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


REL_TOL_Q4_WC = 0.02            # round 6: MLX 4-bit weights on the well-conditioned checkpoint (measured HIP-vs-oracle: see the test's output)


def q4_quantisers(cfg, w):
    """`quantize_model=True, quantize_format="int4"` applied to the oracle's weights: decoder projections + lm_head through the MLX
    group quantiser (weights.mlx_quantize: group 64, 4 bits), kept as the exact fp32 values scale * q + bias -- what
    mx.quantized_matmul multiplies by (phi_3_vision_mlx.py:264,296)."""
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    out = dict(w)
    for k in w:
        if k.startswith("model.layers.") and k.endswith("_proj.weight"):
            out[k] = mlx_dequantize(*mlx_quantize(w[k])).float()
    return out


def q4_wc():
    """`q4_wc_oracle.npz` (round 6): MLX 4-bit group-64 decoder weights on the WELL-CONDITIONED full-size checkpoint, BASELINE config
    1's 128-token text prompt, 16 greedy steps under two unsearched heads (plain, peaked seed 0; both through the quantiser, as the
    build quantises its lm_head).  Pins the round's 4-bit decode kernels at full size -- k_gemv3_q4 with the folded step ends
    (p3v_gemv_q4_step), the merge launch with the 4-bit o_proj (k_attn_combine_o) -- and the dequantise + GEMM prefill against an
    oracle that multiplies by the exact dequantised values."""
    from phi_3_vision_mlx_amd.weights import mlx_dequantize, mlx_quantize
    n_steps = 16
    cfg, o, base = _full_oracle(q4_quantisers, residual_scale=1.0 / 1024)
    ids = np.load(os.path.join(HERE, "c1_oracle.npz"))["ids"]
    r = Prefilled(o, {"input_ids": ids}, n_steps, tag="q4wc")
    r.rel_tol = REL_TOL_Q4_WC

    def q(h):
        return mlx_dequantize(*mlx_quantize(h.to(BF16))).float()
    out = dict(COMMON, rel_tol=np.asarray([REL_TOL_Q4_WC], dtype=np.float32), n_ids=np.asarray([r.S], dtype=np.int32),
               residual_scale=np.asarray([1.0 / 1024], dtype=np.float32), ids=ids)
    for prefix, head in (("plain_", q(base)), ("peaked0_", q(peaked_lm_head(base.to(F32), SPREAD, 0)))):
        res = r.greedy(head, n_steps)
        pack_long(prefix, head, res, out, REL_TOL_Q4_WC)
    np.savez_compressed(os.path.join(HERE, "q4_wc_oracle.npz"), **out)
    print("wrote q4_wc_oracle.npz")


# ---------------------------------------------------------------------------------------------------------------------
# Heavy-tailed activations (weights.add_outliers: 6 residual-stream channels x 64, 2 key / value dimensions per head x 8).
# Three arithmetic variants of config 2's request at FULL size, each against an oracle with the same weights / quantisers:
#   c2h   bf16 weights, bf16 KV                         (BASELINE config 2 under heavy tails)
#   c5wh  e4m3 weights (weight-only), int8 KV           (config 5, `fp8_activations=False`)
#   c5h   e4m3 weights, e4m3 prompt activations (W8A8), int8 KV   (config 5's default path)
# rel_tol of each = 1.3 x the z-space error measured HIP vs oracle (tools/precision_decomp.py hip <tag>,
# profiles/r03_heavy_tail.txt); the tiny twins (2 layers) use the tiny tolerance x the same ratio.
DATA = os.path.join(ROOT, "tools", "data")
HEAVY = {"c2h": dict(quant=False, act8=False), "c5wh": dict(quant=True, act8=False), "c5h": dict(quant=True, act8=True)}
# measured |HIP - oracle| of the residual stream after the last layer (profiles/r03_heavy_tail.txt): full size c2h 3.6-3.9 %,
# c5wh 6.0-7.0 %, c5h 24-28 %; tiny (2 layers) 1.1 % / 1.1 % / 6.8 %.  x 1.3-1.5:
# Round 4: RMSNorm rounds twice on both sides (mx.fast.rms_norm's semantics, pinned by ref_model_tiny.npz).  Under heavy tails the
# extra rounding of the NORMALISED OUTLIER channels (a 1-ulp flip there is a quarter of a typical entry) raised the spread between
# correct implementations -- measured HIP vs oracle per step, two equally correct ViT softmax variants (tools/heavy_diag.py,
# profiles/r04_heavy_tail.txt): c2h 4.7 7.7 3.4 3.8 | 5.7 7.9 3.0 3.5 %; c5wh 3.9 9.0 3.9 5.1 | 4.3 12.8 4.8 4.6 %; c5h 23.8 13.7 64 |
# 20.2 11.8 65 %; tiny c2h <= 2.6, c5wh <= 4.0, c5h <= 4.6 %.  The 7.7-12.8 % steps of c2h / c5wh turned out to be ILL-CONDITIONED
# decode steps (`conditioning` below: they move by > half the tolerance under a bf16 rounding of q / k alone); on heads whose steps
# are all well-conditioned by that test the build measures 4.6-8.5 % (c2h, two heads) and 7.4-8.8 % (c5wh).  Tolerances = 1.2-1.25 x
# the worst of those; c5h (W8A8) 1.25 x its 23.8 %:
REL_TOL_HEAVY = {"c2h": 0.10, "c5wh": 0.13, "c5h": 0.30}     # c5wh: 0.11 until round 5 (its step 1 measured 9.0 .. 12.8 % over equally correct variants; 12.0 % with the rotation's products rounded separately)
REL_TOL_HEAVY_TINY = {"c2h": 0.035, "c5wh": 0.055, "c5h": 0.06}
HEAVY_STEPS = {"c2h": 4, "c5wh": 4, "c5h": 2, "tiny_c2h": 4, "tiny_c5wh": 3, "tiny_c5h": 3}     # c5h: its third step blows up (64 %) since round 4


def heavy(tag, tiny_model=False):
    """One heavy-tail fixture (full size unless tiny_model): prefill + 3 decode steps of config 2's request."""
    from phi_3_vision_mlx_amd.ops import quantize_fp8_rows
    kw = HEAVY[tag]
    if tiny_model:
        cfg = make_config(tiny_config_dict(vision=True))
        w = synth_weights(cfg, seed=0, std_scale=4.0, outliers=True)
        base = w["lm_head.weight"]
        if kw["quant"]:
            w = c5_quantisers(cfg, w)
        o = orc.OraclePhi3V(cfg, w, cache_fp32=True)
    else:
        cfg, o, base = _full_oracle(c5_quantisers if kw["quant"] else None, outliers=True)
    if kw["quant"]:
        o.proj = c5_proj(o, kw["act8"])
    orig = orc.OracleKVCache
    if kw["quant"]:
        orc.OracleKVCache = QuantKVCache
    global peaked_lm_head
    plain = peaked_lm_head
    try:
        inp = vqa_request(Phi3VProcessor(None).img_processor, 0)
        name = ("tiny_" if tiny_model else "") + tag
        n_steps = HEAVY_STEPS[name]
        r = Prefilled(o, {k: (torch.from_numpy(v) if k == "pixel_values" else v) for k, v in inp.items()}, 4,
                      tag=None if tiny_model else name, hidden_file=os.path.join(DATA, f"precision_{name}.npz"))
        r.rel_tol = (REL_TOL_HEAVY_TINY if tiny_model else REL_TOL_HEAVY)[tag]

        def q_head(b, spread, hs):                                    # the peaked head goes through the weight quantiser too
            w8, sc = quantize_fp8_rows(plain(b.to(BF16), spread, hs))
            return w8.view(torch.float8_e4m3fn).to(F32) * sc[:, None]
        if kw["quant"]:
            peaked_lm_head = q_head
        # W8A8 under heavy tails: two correct implementations differ by a quarter of the logit range (every e4m3 activation
        # code that flips is a 6-12 % step of a row whose scale the outlier channels set): no token can be pinned -- the
        # fixture then only bounds the logits (need = 0: whatever head comes first).  The others: first two steps clear.
        first = 0
        while True:
            hs, (res,) = search_head([r], base, n_steps, need=0 if kw["act8"] else 2, min_distinct=1 if kw["act8"] else 2, first_seed=first)
            if kw["act8"]:
                break
            # Round 4: a heavy-tail DECODE step can be ill-conditioned whatever the arithmetic (key dimensions at 8x push some
            # attention logits into the hundreds; a softmax between two competing keys turns a 0.4 % change of a score into tens of
            # per cent at the logits).  Such a step is not fixture material for ANY correct implementation, and which steps they are
            # depends on the tokens the head picks.  Objective, GPU-independent test: replay the fixture's steps with the rotated
            # q / k rounded to bf16 (the one rounding every bf16 attention makes) and require every step to move by < half the
            # tolerance; else take the next head.
            cond = conditioning(r, peaked_lm_head(base.to(F32), SPREAD, hs), res[0], n_steps)
            print(f"  head {hs}: conditioning per step (z-space change under bf16 q / k, x tolerance) {[round(c / r.rel_tol, 2) for c in cond]}", flush=True)
            if max(cond) < 0.5 * r.rel_tol:
                break
            first = hs + 1
    finally:
        peaked_lm_head = plain
        orc.OracleKVCache = orig
    out = dict(COMMON, rel_tol=np.asarray([r.rel_tol], dtype=np.float32), n_ids=np.asarray([r.S], dtype=np.int32))
    pack("", hs, res, out)
    np.savez_compressed(os.path.join(HERE, f"{name}_oracle.npz"), **out)
    print("wrote", f"{name}_oracle.npz")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which.startswith("heavy:"):                              # heavy:c2h | heavy:tiny_c5h | ...
        t = which.split(":", 1)[1]
        heavy(t.replace("tiny_", ""), tiny_model=t.startswith("tiny_"))
        sys.exit(0)
    if which in ("tiny", "all"):
        tiny()
    if which in ("tiny_serve", "all"):
        tiny_serve()
    if which in ("full", "all"):
        full()
    if which in ("c3", "all"):
        c3()
    if which in ("c5", "all"):
        c5(True)
    if which in ("c5w", "all"):
        c5(False)
    if which == "c5_wc":
        c5_wc(True)
    if which == "c5w_wc":
        c5_wc(False)
    if which == "q4_wc":
        q4_wc()
