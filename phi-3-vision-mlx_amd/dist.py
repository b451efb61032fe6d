"""Batch sharding over the GPUs of one node.

The reference has no distributed code; sequences never interact (per-row mask and
position ids, reference phi.py:238-239, 553-559), so the path shards by REQUEST:
one process per GPU (`torchrun`), replicated weights, rank r serves requests
``r, r+W, r+2W, ...``.  RCCL (``backend="nccl"``) over xGMI is used only at request
boundaries -- the request table, an optional one-time weight broadcast, and the
gather of results -- kilobytes per call, never per token, and always as TENSORS on the
collective's own device (no pickled object crosses RCCL); there is no tensor
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


def _coll_device(group=None):
    """Where collective payloads live: the current GPU under RCCL (backend "nccl"), host memory under gloo."""
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def pack_requests(prompts, images):
    """(prompts, images) -> (table int64 [3 + n + 3 * n_img], payload uint8): the request table as two flat tensors.
    images[i]: None | one image | a list of images, each a PIL image / path / URL as `generate` takes them (decoded HERE,
    on the rank that holds the table, to RGB uint8 pixels).  table = [n, n_img, text_bytes | per prompt: utf-8 length |
    per image: prompt index, height, width]; payload = the prompts' utf-8 bytes, then every image's H x W x 3 pixels."""
    from .api import _load_image
    n = len(prompts)
    images = list(images) if images is not None else [None] * n
    texts = [p.encode("utf-8") for p in prompts]
    rows, pix = [], []
    for i, entry in enumerate(images):
        for im in ([] if entry is None else entry if isinstance(entry, (list, tuple)) else [entry]):
            a = np.ascontiguousarray(np.asarray(_load_image(im).convert("RGB"), dtype=np.uint8))
            rows += [i, a.shape[0], a.shape[1]]
            pix.append(a.reshape(-1))
    table = np.asarray([n, len(rows) // 3, sum(len(t) for t in texts)] + [len(t) for t in texts] + rows, dtype=np.int64)
    payload = np.concatenate([np.frombuffer(b"".join(texts), dtype=np.uint8)] + pix) if (texts or pix) else np.zeros(0, np.uint8)
    return torch.from_numpy(table), torch.from_numpy(payload.copy())


def unpack_requests(table, payload):
    """Inverse of `pack_requests`: (prompts, images) with images[i] = None | PIL image | list of PIL images."""
    from PIL import Image
    t, buf = table.cpu().numpy(), payload.cpu().numpy()
    n, n_img, text_bytes = (int(v) for v in t[:3])
    lens, rows = t[3:3 + n], t[3 + n:3 + n + 3 * n_img].reshape(n_img, 3)
    prompts, off = [], 0
    for ln in lens:
        prompts.append(bytes(buf[off:off + int(ln)]).decode("utf-8"))
        off += int(ln)
    assert off == text_bytes
    per = [[] for _ in range(n)]
    for i, h, w in rows:
        k = int(h) * int(w) * 3
        per[int(i)].append(Image.fromarray(buf[off:off + k].reshape(int(h), int(w), 3).copy(), "RGB"))
        off += k
    images = [None if not p else p[0] if len(p) == 1 else p for p in per]
    return prompts, (images if n_img else None)


def broadcast_requests(prompts, images, src=0, group=None):
    """Rank `src`'s request table becomes everybody's: three broadcasts -- the two tensor sizes, the int64 table, the uint8
    payload (prompt text + decoded RGB pixels) -- on the collective's own device (RCCL: device tensors over xGMI; no pickle,
    no PIL object on the wire).  A 336 x 336 image is 339 KB; config 4's 32 images ~ 11 MB, once per call."""
    rank, world = _world(group)
    if world == 1:
        return list(prompts), (list(images) if images is not None else None)
    dev = _coll_device(group)
    if rank == src:
        table, payload = pack_requests(prompts, images)
        sizes = torch.tensor([table.numel(), payload.numel()], dtype=torch.int64)
    else:
        table = payload = None
        sizes = torch.zeros(2, dtype=torch.int64)
    sizes = sizes.to(dev)
    dist.broadcast(sizes, src=src, group=group)
    n_t, n_p = (int(v) for v in sizes.cpu().tolist())
    table = table.to(dev) if rank == src else torch.empty(n_t, dtype=torch.int64, device=dev)
    payload = payload.to(dev) if rank == src else torch.empty(n_p, dtype=torch.uint8, device=dev)
    dist.broadcast(table, src=src, group=group)
    if n_p:
        dist.broadcast(payload, src=src, group=group)
    return unpack_requests(table, payload)


def _all_gather_ragged(t, group=None):
    """1-D tensors of different lengths, one per rank -> list of them (two tensor collectives: the lengths, then the data padded
    to the longest; on the collective's own device)."""
    rank, world = _world(group)
    dev = _coll_device(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(v.item()) for v in ns]
    width = max(max(ns), 1)
    mine = torch.zeros(width, dtype=t.dtype, device=dev)
    mine[:t.numel()] = t.to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return [b[:k].cpu() for b, k in zip(bufs, ns)]


_K_TEXT, _K_INTS, _K_FLOATS, _K_INT, _K_FLOAT = 0, 1, 2, 3, 4


def _encode_result(r):
    """One result -> (kind, int32 words).  Texts (utf-8 bytes, one per word), integer sequences / scalars, float sequences / scalars
    (float64 bit patterns, two words each: nothing is truncated); anything else raises instead of being coerced."""
    import numbers
    import struct
    if isinstance(r, str):
        return _K_TEXT, list(r.encode("utf-8"))
    if torch.is_tensor(r):
        r = r.tolist()
    elif hasattr(r, "tolist") and not isinstance(r, (list, tuple)):
        r = r.tolist()                                              # numpy arrays / scalars
    scalar = not isinstance(r, (list, tuple))
    vals = [r] if scalar else list(r)
    if all(isinstance(v, numbers.Integral) and not isinstance(v, bool) for v in vals):
        if any(not -2 ** 31 <= int(v) < 2 ** 31 for v in vals):
            raise TypeError("gather_results: integer results must fit int32")
        return (_K_INT if scalar else _K_INTS), [int(v) for v in vals]
    if all(isinstance(v, numbers.Real) and not isinstance(v, bool) for v in vals):
        words = []
        for v in vals:
            words += struct.unpack("<ii", struct.pack("<d", float(v)))
        return (_K_FLOAT if scalar else _K_FLOATS), words
    raise TypeError(f"gather_results carries texts, integer or real sequences and scalars, not {type(r).__name__} of "
                    f"{sorted({type(v).__name__ for v in vals})}")


def _decode_result(kind, words):
    import struct
    if kind == _K_TEXT:
        return bytes(words).decode("utf-8")
    if kind in (_K_INTS, _K_INT):
        return words[0] if kind == _K_INT else words
    vals = [struct.unpack("<d", struct.pack("<ii", words[i], words[i + 1]))[0] for i in range(0, len(words), 2)]
    return vals[0] if kind == _K_FLOAT else vals


def pack_results(local_idx, local_results):
    """This rank's results as ONE int32 tensor: [n | request ids | kinds | lengths | payload] (kinds: `_encode_result`).
    (Round 5: the result gather used to pickle Python objects through all_gather_object; round 6: typed per result, so float
    scores and scalar results survive a world > 1 gather unchanged instead of being truncated / raising.)"""
    enc = [_encode_result(r) for r in local_results]
    flat = [len(enc)] + [int(i) for i in local_idx] + [k for k, _ in enc] + [len(q) for _, q in enc] + [v for _, q in enc for v in q]
    return torch.tensor(flat, dtype=torch.int32)


def unpack_results(t, out):
    v = t.tolist()
    n = v[0]
    idx, kinds, lens, off = v[1:1 + n], v[1 + n:1 + 2 * n], v[1 + 2 * n:1 + 3 * n], 1 + 3 * n
    for i, kind, ln in zip(idx, kinds, lens):
        out[i] = _decode_result(kind, v[off:off + ln])
        off += ln


def gather_results(local_idx, local_results, n_total, group=None):
    """All ranks end up with the full, request-ordered result list (texts or token lists): tensor collectives only."""
    rank, world = _world(group)
    out = [None] * n_total
    if world == 1:
        for i, r in zip(local_idx, local_results):
            out[i] = _decode_result(*_encode_result(r))           # (the same validation and value types as a world > 1 gather)
        return out
    for t in _all_gather_ragged(pack_results(local_idx, local_results), group):
        unpack_results(t, out)
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
    """`worker(indices) -> list of results` on this rank's shard; returns the gathered, ordered list.  A result is a text, a sequence
    of integers (int32) or reals, or an integer / real scalar (`pack_results`); anything else raises TypeError on every world size."""
    rank, world = _world(group)
    idx = shard_indices(n_requests, rank, world)
    local = worker(idx) if idx else []
    return gather_results(idx, local, n_requests, group)


def _all_max(value, group=None):
    """max of a Python int over all ranks (request-boundary collective, a few bytes)."""
    rank, world = _world(group)
    if world == 1:
        return value
    t = torch.tensor([int(value)], dtype=torch.int64, device=_coll_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def generate_sharded(prompts, images=None, preload=None, max_tokens=512, group=None, max_batch=8, apply_chat_template=True,
                     return_tokens=False):
    """Batched `generate()` over all ranks -- BASELINE config 4 (mixed image + text requests).

    `prompts`: list of str.  `images`: None, or a list with one entry (an image / path / URL as `generate` takes it, or
    None) per prompt.  Rank 0's request table is broadcast (`broadcast_requests`: text + decoded RGB pixels as one uint8 tensor, ~340 KB per 336x336 image);
    rank r serves requests r, r+W, ... in chunks of `max_batch` rows.  A chunk -- image and text requests alike -- runs as
    ONE left-padded batch (`processor.collate_requests`: batched ViT, one prefill, graph-replayed batched decode), padded
    to the longest prompt of the WHOLE request list so that the pad geometry and the one-shot short / long RoPE choice
    (phi.py:492, Q2) do not depend on the number of ranks: sharded output == single-process output, request-ordered.
    The reference runs image prompts at B = 1 only (phi_3_vision_mlx.py:377-378); per row the batch computes what the
    B = 1 run computes (tests/test_model_gpu.py::test_c4_share_batched_vs_per_request_oracle).
    Returns the full list of texts (or token lists) on every rank."""
    from . import api
    from .processor import collate_requests
    rank, world = _world(group)
    prompts, images = broadcast_requests(list(prompts), list(images) if images is not None else None, group=group)
    n = len(prompts)
    images = images if images is not None else [None] * n
    model, processor = preload
    idx = shard_indices(n, rank, world)
    reqs = []
    for i in idx:
        if apply_chat_template:
            text, imgs = api._apply_chat_template(prompts[i], images[i], False)
        else:
            text, imgs = prompts[i], (None if images[i] is None else [api._load_image(images[i])])
        reqs.append(processor(text, imgs) if imgs is not None else processor(text))
    width = _all_max(max([np.asarray(r["input_ids"]).shape[-1] for r in reqs], default=0), group)
    local = []
    for c in range(0, len(reqs), max_batch):
        local += generate_requests(model, processor, reqs[c:c + max_batch], max_tokens, return_tokens, width=width)
    return gather_results(idx, local, n, group)


class _SlotCache:
    def __init__(self, state):
        self.state = state


GROUP_PAD = 256          # prefill_requests: rows of padding a request may carry inside its length group


def prefill_requests(model, reqs, max_tokens, width=None):
    """Prefill B = 1 requests of DIFFERENT lengths into one decode batch without computing on padding.

    A left-padded batch (`collate_requests`) runs every projection over B x longest-prompt rows: for one GPU's share of
    config 4 (4 x 2531-token image prompts + 4 text prompts of 65..233 tokens) almost half of them are padding.  Here the
    requests are sorted by length, each group of equal (or, up to GROUP_PAD pad rows, nearly equal) length is prefilled as its own batch straight into adjacent rows of a
    slot state (`model.prefill_slot`: right-aligned to the longest prompt, per-row left padding and position tables -- the
    geometry `_tokenize` gives a padded row, phi.py:238-240), and decode then runs all rows as ONE graph-replayed batch.
    The slot window is W + max_tokens with W the longest prompt of the WHOLE request list, so the rows get the RoPE factors
    the padded batch would pick (phi.py:492; short up to 4096, long beyond) and, with `quantize_cache=True`, the int8 cache.
    Returns (first tokens int32 [B, 1] in SLOT order, cache, order) with order[slot] = request index; None when the
    model has no slot states (callers then take the padded batch)."""
    lens = [int(np.asarray(r["input_ids"]).shape[-1]) for r in reqs]
    W = max(max(lens), width or 0)
    if not hasattr(model, "new_slot_state"):
        return None
    from .processor import collate_requests
    order = sorted(range(len(reqs)), key=lambda i: (-lens[i], i))
    st = model.new_slot_state(len(reqs), W + max_tokens)
    st.offset = W
    g = model.decode_graph(st)
    row = 0
    while row < len(order):
        # a group = requests of equal length, plus shorter ones whose padding inside the group stays small: a short prompt
        # prefilled alone still streams all 7.4 GB of weights (config 4's four text prompts of 65..233 tokens cost 9.6 ms EACH
        # that way); padded to the group's longest they share one pass (<= GROUP_PAD pad rows per request)
        n, longest = 1, lens[order[row]]
        while row + n < len(order) and longest - lens[order[row + n]] <= GROUP_PAD:
            n += 1
        group = [reqs[i] for i in order[row:row + n]]
        tok = model.prefill_slot(st, row, collate_requests(group) if n > 1 else group[0])
        g["tok"][row:row + n].copy_(tok.reshape(-1))
        row += n
    return g["tok"].view(-1, 1), [_SlotCache(st)], order


def generate_requests(model, processor, reqs, max_tokens, return_tokens=False, width=None):
    """Greedy generation for a list of B = 1 model inputs as one decode batch; results in request order."""
    from . import api
    from .processor import collate_requests
    pre = prefill_requests(model, reqs, max_tokens, width)
    if pre is None:
        return generate_rows(model, processor, collate_requests(reqs, width=width), max_tokens, return_tokens)
    token, cache, order = pre
    streamer = api.Streamer(processor, False, True)
    stopper = api.TokenStopper(processor, len(reqs))
    streamer(api._rows(token))
    api.greedy_loop(model, token, cache, max_tokens - 1, streamer, stopper)
    if return_tokens:
        per_slot = [list(r) for r in zip(*streamer.list_tokens)]
        per_slot = [(r[:r.index(api.ID_EOS) + 1] if api.ID_EOS in r else r) for r in per_slot]
    else:
        per_slot = list(streamer.end()[0])
    out = [None] * len(reqs)
    for slot, i in enumerate(order):
        out[i] = per_slot[slot]
    return out


def generate_rows(model, processor, rows, max_tokens, return_tokens=False):
    """Greedy loop of `_generate` (phi_3_vision_mlx.py:384-400) on a collated batch; EOS-trimmed texts (or token lists)."""
    from . import api
    token, cache = model.greedy_prefill(max_tokens, **rows)
    streamer = api.Streamer(processor, False, True)
    stopper = api.TokenStopper(processor, np.asarray(rows["input_ids"]).shape[0])
    streamer(api._rows(token))
    api.greedy_loop(model, token, cache, max_tokens - 1, streamer, stopper)
    if return_tokens:
        per_row = [list(r) for r in zip(*streamer.list_tokens)]
        return [(r[:r.index(api.ID_EOS) + 1] if api.ID_EOS in r else r) for r in per_row]
    return list(streamer.end()[0])
