"""Batch sharding over the GPUs of one node.

The reference has no distributed code; sequences never interact (per-row mask and
position ids, reference phi.py:238-239, 553-559), so the path shards by REQUEST:
one process per GPU (`torchrun`), replicated weights, rank r serves requests
``r, r+W, r+2W, ...``.  RCCL (``backend="nccl"``) over xGMI is used only at request
boundaries -- the request table, an optional one-time weight broadcast, and the
gather of results -- kilobytes per call, never per token; there is no tensor
parallelism.  Every helper also runs on gloo/CPU (tests/test_dist_cpu.py).

Left-pad geometry and the one-shot short/long RoPE choice (Q2) depend on the
longest prompt of the WHOLE batch, so each rank tokenises the full prompt list
(host work, microseconds) and keeps its rows: the shard is bit-identical to the
corresponding rows of the single-GPU batch.
"""
import numpy as np
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_indices(n, rank, world):
    """Request ids served by `rank` (round-robin: balances mixed long/short requests)."""
    return list(range(rank, n, world))


def shard_text_batch(processor, prompts, rank, world):
    """Rows of the global left-padded batch (`Phi3FProcessor._tokenize`, phi.py:233-245) owned by `rank`."""
    idx = shard_indices(len(prompts), rank, world)
    full = processor(list(prompts))
    return idx, {k: np.ascontiguousarray(np.asarray(v)[idx]) for k, v in full.items()}


def broadcast_requests(requests, src=0, group=None):
    """Rank `src`'s request table becomes everybody's (ncclBroadcast of a pickled blob)."""
    rank, world = _world(group)
    if world == 1:
        return requests
    box = [requests if rank == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def gather_results(local_idx, local_results, n_total, group=None):
    """All ranks end up with the full, request-ordered result list."""
    rank, world = _world(group)
    if world == 1:
        out = [None] * n_total
        for i, r in zip(local_idx, local_results):
            out[i] = r
        return out
    boxes = [None] * world
    dist.all_gather_object(boxes, (list(local_idx), list(local_results)), group=group)
    out = [None] * n_total
    for idx, res in boxes:
        for i, r in zip(idx, res):
            out[i] = r
    return out


def gather_tokens(tokens, group=None):
    """[B_local, T] int32 device tensor per rank -> [W, B_local, T] on every rank (ncclAllGather)."""
    rank, world = _world(group)
    if world == 1:
        return tokens[None]
    bufs = [torch.empty_like(tokens) for _ in range(world)]
    dist.all_gather(bufs, tokens.contiguous(), group=group)
    return torch.stack(bufs)


def sync_weights(weights, src=0, group=None):
    """One-time replication of rank `src`'s weights (7.6 GB bf16 ~ 50 ms per xGMI link-bound ring)."""
    rank, world = _world(group)
    if world == 1:
        return weights
    for k in sorted(weights):
        dist.broadcast(weights[k], src=src, group=group)
    return weights


def run_sharded(n_requests, worker, group=None):
    """`worker(indices) -> list of results` on this rank's shard; returns the gathered, ordered list."""
    rank, world = _world(group)
    idx = shard_indices(n_requests, rank, world)
    local = worker(idx) if idx else []
    return gather_results(idx, local, n_requests, group)


def generate_sharded(prompts, images=None, preload=None, max_tokens=512, group=None, **kwargs):
    """Batched `generate()` over all ranks.

    `prompts`: list of str.  `images`: None, or a list (one entry or None per prompt).
    Text-only requests of a rank run as ONE left-padded batch padded to the global
    maximum length; image requests run one by one (the reference only supports B=1
    with images, phi_3_vision_mlx.py:377-378).  Returns the full list on every rank."""
    from . import api
    rank, world = _world(group)
    prompts = broadcast_requests(list(prompts), group=group)
    n = len(prompts)
    images = list(images) if images is not None else [None] * n
    model, processor = preload

    def worker(idx):
        out = {}
        text_idx = [i for i in idx if images[i] is None]
        if text_idx:
            templ, _ = api._apply_chat_template([prompts[i] for i in range(n)], None, False)
            templ = [templ] if isinstance(templ, str) else templ
            full = processor(templ)
            rows = {k: np.ascontiguousarray(np.asarray(v)[text_idx]) for k, v in full.items()}
            texts = _generate_rows(model, processor, rows, max_tokens)
            out.update(dict(zip(text_idx, texts)))
        for i in idx:
            if images[i] is not None:
                r = api.generate(prompts[i], images[i], preload=preload, max_tokens=max_tokens, verbose=False, stream=False, **kwargs)
                out[i] = r[0] if isinstance(r, list) else r
        return [out[i] for i in idx]
    return run_sharded(n, worker, group)


def _generate_rows(model, processor, rows, max_tokens):
    """Greedy loop of `_generate` (phi_3_vision_mlx.py:384-400) on pre-tokenised rows; EOS-trimmed texts."""
    from . import api, ops
    logits, cache = model(**rows, max_tokens=max_tokens)
    token = ops.argmax(logits[:, -1, :].contiguous())[:, None]
    streamer = api.Streamer(processor, False, True)
    stopper = api.TokenStopper(processor, rows["input_ids"].shape[0])
    streamer(token)
    for _ in range(max_tokens - 1):
        logits, token = model.greedy_step(token, cache)
        streamer(token)
        if stopper(token):
            break
    return streamer.end()[0]
