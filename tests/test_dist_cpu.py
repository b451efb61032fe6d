"""Batch sharding over ranks, world_size 2 on gloo/CPU (the GPU box runs the same code on RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phi_3_vision_mlx_amd import dist as pd
    from phi_3_vision_mlx_amd.processor import Phi3FProcessor
    prompts = [f"prompt number {i} " + "x" * (3 * i) for i in range(7)]
    from golden_inputs import make_image
    imgs = [make_image(336, 336, "noise", 0), None, [make_image(64, 48, "smooth", 1), make_image(20, 30, "noise", 2)]] + [None] * 4
    got_p, got_i = pd.broadcast_requests(*((prompts, imgs) if rank == 0 else (["garbage"], None)))
    assert got_p == prompts and got_i[1] is None and got_i[3] is None and len(got_i) == 7
    assert np.array_equal(np.asarray(got_i[0]), np.asarray(imgs[0])) and got_i[0].mode == "RGB"       # pixels, not pickles
    assert isinstance(got_i[2], list) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got_i[2], imgs[2]))
    got_p, got_i = pd.broadcast_requests(*((prompts[:2] + ["unicode \u00e9\u4e2d"], None) if rank == 0 else ([], None)))
    assert got_p == prompts[:2] + ["unicode \u00e9\u4e2d"] and got_i is None
    # sharded work + ordered gather
    res = pd.run_sharded(len(prompts), lambda idx: [prompts[i].upper() for i in idx])
    assert res == [p.upper() for p in prompts]
    # shard rows are bit-identical to the rows of the global left-padded batch
    proc = Phi3FProcessor(None)
    idx, rows = pd.shard_text_batch(proc, prompts, rank, world)
    full = proc(prompts)
    assert idx == list(range(rank, 7, world))
    for k in ("input_ids", "pids", "mask"):
        assert np.array_equal(rows[k], np.asarray(full[k])[idx])
    assert rows["input_ids"].shape[1] == np.asarray(full["input_ids"]).shape[1]      # padded to the GLOBAL max
    # token gather + weight replication
    toks = torch.full((2, 5), rank, dtype=torch.int32)
    allt = pd.gather_tokens(toks)
    assert allt.shape == (world, 2, 5) and all((allt[r] == r).all() for r in range(world))
    w = {"a": torch.full((4,), float(rank)), "b": torch.arange(3, dtype=torch.float32) * (rank + 1)}
    pd.sync_weights(w, src=0)
    assert (w["a"] == 0).all() and torch.equal(w["b"], torch.arange(3, dtype=torch.float32))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


class StubModel:
    """CPU stand-in with the engine's interface (`greedy_prefill` / `greedy_step`): each row's tokens are a function of
    that row's valid prompt ids, its image (if any) and the step only -- like the real model, pad-invariant."""
    device = "cpu"

    def greedy_prefill(self, max_tokens, input_ids, pids=None, mask=None, pixel_values=None, image_sizes=None, positions=None):
        ids, m = np.asarray(input_ids), np.asarray(mask)
        assert np.array_equal(np.asarray(pids), np.where(m == 1, np.cumsum(m, axis=1) - 1, 1))     # _tokenize's conventions
        key = (np.where(ids > 0, ids, 0) * m).sum(axis=1)
        if pixel_values is not None:
            pos = np.asarray(positions)
            rows = pos[np.concatenate([[True], pos[1:, 0] != pos[:-1, 0]]), 0]
            assert (ids[pos[:, 0], pos[:, 1]] < 0).all()                                  # slots moved with the left pad
            for k, r in enumerate(rows):
                key[r] += int(np.abs(np.asarray(pixel_values[k], dtype=np.float64)).sum()) % 9973
        self.key, self.step = key, 0
        return self._tok(), [self]

    def _tok(self):
        t = (self.key * 31 + self.step * 7919) % 31000 + 3
        t = np.where((self.key + self.step) % 11 == 10, 32007, t)                          # some rows stop early (EOS)
        return torch.as_tensor(t[:, None].astype(np.int32))

    def greedy_step(self, token, cache):
        self.step += 1
        return None, self._tok()


def _worker_generate(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_inputs import make_image
    from phi_3_vision_mlx_amd import api, dist as pd
    from phi_3_vision_mlx_amd.processor import Phi3VProcessor, collate_requests
    proc = Phi3VProcessor(None, return_mx=False)
    prompts = [f"question {i} " + "y" * (5 * i) for i in range(7)]
    images = [make_image(336, 336, "noise", 3), None, None, make_image(640, 480, "smooth", 1), None, None, None]
    mine = (prompts, images) if rank == 0 else (["junk"], None)          # only rank 0 holds the request table

    def no_pickle(*a, **k):                                               # (round 5) nothing on this path may pickle objects through a collective
        raise AssertionError("object collective used: the batch-sharded path moves tensors only")
    dist.all_gather_object = dist.broadcast_object_list = dist.gather_object = no_pickle
    got = pd.generate_sharded(*mine, preload=(StubModel(), proc), max_tokens=6, max_batch=2, return_tokens=True)
    # single-process result: every request collated into one batch
    reqs = []
    for p_, im in zip(prompts, images):
        text, imgs = api._apply_chat_template(p_, im, False)
        reqs.append(proc(text, imgs) if imgs is not None else proc(text))
    want = pd.generate_rows(StubModel(), proc, collate_requests(reqs), 6, return_tokens=True)
    assert got == want and len(got) == 7, (got, want)
    assert any(r[-1] == 32007 and len(r) < 6 for r in got) and any(len(r) == 6 for r in got)
    texts = pd.generate_sharded(*mine, preload=(StubModel(), proc), max_tokens=4)
    assert isinstance(texts, list) and len(texts) == 7 and all(isinstance(t, str) for t in texts)
    uni = pd.gather_results([rank], ["r\u00e9sum\u00e9 \u2713 %d" % rank], world)      # non-ASCII text survives the byte packing
    assert uni == ["r\u00e9sum\u00e9 \u2713 0", "r\u00e9sum\u00e9 \u2713 1"], uni
    # typed results: float scores are not truncated, scalar results and mixed kinds survive a world-2 gather
    assert pd.run_sharded(5, lambda idx: [i * i for i in idx]) == [0, 1, 4, 9, 16]
    assert pd.run_sharded(4, lambda idx: [[0.1 * i, -1e-30, 3.0e300] for i in idx]) == [[0.1 * i, -1e-30, 3.0e300] for i in range(4)]
    mixed = pd.gather_results([rank, rank + 2], [0.5 + rank, "t%d" % rank] if rank == 0 else [[1, -2, 3], 7], 4)
    assert mixed == [0.5, [1, -2, 3], "t0", 7], mixed
    for bad in ([object()], [[1, "a"]], [2 ** 40], [{"a": 1}]):
        try:
            pd.pack_results([0], bad)
            raise AssertionError(f"pack_results accepted {bad!r}")
        except TypeError:
            pass
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"gen{rank}"), "w").write("ok")


def test_generate_sharded_world2_equals_single_process(tmp_path):
    """`generate_sharded` (mixed image + text requests, rank 0 holds the table, chunks of 2 rows per rank) on 2 gloo ranks
    == the single-process result, request-ordered, EOS-trimmed."""
    world, port = 2, _free_port()
    mp.spawn(_worker_generate, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"gen{r}") for r in range(world))


class FastImageProcessor:
    """The image stage is not what this file tests (tests/test_processor_golden.py pins it to the reference): a stand-in with
    the real OUTPUT GEOMETRY of a 336x336 request (17 crops, 2509 image tokens -> 2531-token prompts)
    whose pixel content is a cheap function of the image, so 64-request tables run in seconds on the CPU."""

    def __call__(self, images, dtype=None):
        px = [np.asarray(im, dtype=np.float64).mean(axis=(0, 1)) for im in images]
        pv = np.stack([np.broadcast_to(p[None, :, None, None], (17, 3, 2, 2)) for p in px]).copy()
        return {"pixel_values": pv, "image_sizes": [[1344, 1344]] * len(images), "num_img_tokens": [2509] * len(images)}


def config4_table(n):
    """BASELINE config 4's request table: n/2 single-image VQA prompts + n/2 text prompts of length ~U[16, 256]."""
    from golden_inputs import make_image
    rng = np.random.default_rng(4)
    prompts, images = [], []
    for i in range(n):
        if (i // 8) % 2 == 0:                                # r::8 sharding then gives every rank images AND texts
            prompts.append(f"<|image_1|>\nWhat is in image {i}?")
            images.append(make_image(32, 32, "noise", i))
        else:
            prompts.append("t" * int(rng.integers(16, 257)))
            images.append(None)
    return prompts, images


def _worker_config4(rank, world, port, out_dir, n):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests", "golden")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phi_3_vision_mlx_amd import api, dist as pd
    from phi_3_vision_mlx_amd.processor import Phi3VProcessor, collate_requests
    proc = Phi3VProcessor(None, return_mx=False)
    proc.img_processor = FastImageProcessor()
    prompts, images = config4_table(n)
    mine = (prompts, images) if rank == 0 else (["junk"], None)          # only rank 0 holds the request table
    got = pd.generate_sharded(*mine, preload=(StubModel(), proc), max_tokens=6, max_batch=8, return_tokens=True)
    assert len(got) == n and all(g is not None for g in got)
    if rank == 0:                                                        # the single-process result: ONE batch of all requests
        reqs = []
        for p_, im in zip(prompts, images):
            text, imgs = api._apply_chat_template(p_, im, False)
            reqs.append(proc(text, imgs) if imgs is not None else proc(text))
        lens = [np.asarray(r["input_ids"]).shape[-1] for r in reqs]
        assert max(lens) > 2500 and min(lens) < 300                      # mixed image / text lengths, as config 4
        want = pd.generate_rows(StubModel(), proc, collate_requests(reqs), 6, return_tokens=True)
        assert got == want, [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:3]
        assert len({tuple(g) for g in got}) > n // 2                     # not a constant
    boxes = [None] * world
    dist.all_gather_object(boxes, got)
    assert all(b == got for b in boxes)                                  # every rank ends up with the same full list
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"c4_{rank}"), "w").write("ok")


@pytest.mark.parametrize("n", [64, 13])
def test_config4_table_over_8_ranks_equals_single_process(tmp_path, n):
    """BASELINE config 4 as the scaling bench runs it: 64 mixed requests split r::8 over EIGHT ranks (8 per rank, uneven
    prompt lengths 2531 vs 16..256) -- and 13 requests, so that some ranks serve 2 and some 1 -- with rank 0 alone
    holding the table: request-ordered output on every rank == the single-process batch."""
    world, port = 8, _free_port()
    mp.spawn(_worker_config4, args=(world, port, str(tmp_path), n), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"c4_{r}") for r in range(world))


def test_sharding_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_single_process_fallbacks():
    from phi_3_vision_mlx_amd import dist as pd
    table, payload = pd.pack_requests(["a", "bc"], [None, None])
    assert table.dtype == torch.int64 and payload.dtype == torch.uint8 and pd.unpack_requests(table, payload) == (["a", "bc"], None)
    assert pd.broadcast_requests(["x"], None) == (["x"], None)
    assert pd.shard_indices(10, 1, 4) == [1, 5, 9]
    assert pd.run_sharded(3, lambda idx: [i * i for i in idx]) == [0, 1, 4]
    t = torch.zeros(2, 3, dtype=torch.int32)
    assert pd.gather_tokens(t).shape == (1, 2, 3)
