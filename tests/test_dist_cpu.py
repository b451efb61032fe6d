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
    got = pd.broadcast_requests(prompts if rank == 0 else ["garbage"])
    assert got == prompts
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


def test_sharding_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_single_process_fallbacks():
    from phi_3_vision_mlx_amd import dist as pd
    assert pd.shard_indices(10, 1, 4) == [1, 5, 9]
    assert pd.run_sharded(3, lambda idx: [i * i for i in idx]) == [0, 1, 4]
    t = torch.zeros(2, 3, dtype=torch.int32)
    assert pd.gather_tokens(t).shape == (1, 2, 3)
