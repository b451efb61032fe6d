"""fleet.py on CPU: three gloo ranks, each stepping its own continuous-batching engine over the stand-in model of
tests/test_engine_cpu.py (a row's tokens are a function of its own prompt and step, as with the real model).  Rank 0 dispatches;
every request's tokens must equal what ONE local engine produces for it, whichever rank ran it -- plus load spreading,
remote failures arriving as the right exception type, cancellation, a clean shutdown.  (Real kernels: tests/test_serving_gpu.py.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _req(n, seed):
    return {"input_ids": np.random.default_rng(seed).integers(3, 30000, (1, n)).astype(np.int64)}


def _worker(rank, world, port, out_dir):
    import sys
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_engine_cpu import SlotStub
    from phi_3_vision_mlx_amd import fleet
    from phi_3_vision_mlx_amd.engine import ContinuousEngine
    eng = ContinuousEngine(SlotStub(), None, slots=2, window=4096)
    import time
    assert fleet.IDLE_TIMEOUT_DAYS >= 365      # blocking recvs wait for the next request -- they must never expire
    groups = fleet.make_groups()               # (gloo closes the pair on a recv timeout: it cannot be caught and retried)
    if rank:
        fleet.worker(eng, groups)
        open(os.path.join(out_dir, f"worker{rank}"), "w").write("ok")
        dist.destroy_process_group()
        return
    front = fleet.EngineFleet(eng, groups, world)
    stop = threading.Event()
    stepper = threading.Thread(target=front.serve_forever, args=(stop,), daemon=True)
    stepper.start()
    # 9 requests of different lengths and budgets, submitted at once: spread over the ranks by outstanding load (ties to the
    # lowest rank; a rank that finishes early gets more)
    time.sleep(1.0)          # an idle spell before the first request
    shapes = [(20 + 7 * i, 5 + (i % 4)) for i in range(9)]
    hs = [front.submit(_req(n, 100 + i), m) for i, (n, m) in enumerate(shapes)]
    assert all(h.done.wait(60) for h in hs) and all(h.error is None for h in hs), [h.error for h in hs]
    assert sum(front.sent) == 9 and min(front.sent) >= 1, front.sent
    # reference: one local engine, one request at a time
    ref_eng = ContinuousEngine(SlotStub(), None, slots=2, window=4096)
    for i, ((n, m), h) in enumerate(zip(shapes, hs)):
        r = ref_eng.submit(_req(n, 100 + i), m)
        ref_eng.run_until_idle()
        assert r.tokens == h.tokens and 1 <= len(h.tokens) <= m, (i, r.tokens, h.tokens)
    # a request the remote model rejects (token 666 -> ValueError in prefill) fails alone, with the exception's type
    from phi_3_vision_mlx_amd.engine import Request
    busy = [Request(_req(5, 0), 1) for _ in range(5)]                   # handles that never finish: rank 0 looks loaded ...
    front.local, front.load[2] = list(busy), 5                          # ... and so does rank 2: the next ones go to rank 1
    bad = _req(30, 7)
    bad["input_ids"][0, 3] = 666
    hb, hg = front.submit(bad, 4), front.submit(_req(33, 8), 4)         # a good one right behind the bad one
    assert hb.rank == 1 and hg.rank == 1
    assert hb.done.wait(60) and isinstance(hb.error, ValueError) and "rank 1" in str(hb.error)
    assert hg.done.wait(60) and hg.error is None and len(hg.tokens) >= 1
    # cancellation reaches the remote engine: a long request is dropped, its slot serves the next ones
    front.local, front.load[2] = list(busy) + list(busy), 9             # (still steering everything to rank 1)
    long_h = front.submit(_req(40, 11), 3000)
    assert long_h.rank == 1
    long_h.cancel()
    nxt = [front.submit(_req(25 + i, 12 + i), 4) for i in range(3)]     # rank 1 has 2 slots: the third needs the cancelled one's
    assert all(h.done.wait(60) and h.error is None for h in nxt)
    front.local, front.load[2] = [], 0
    front.close()
    stop.set()
    stepper.join(5)
    assert not any(t.is_alive() for t in front.receivers)
    open(os.path.join(out_dir, "front"), "w").write("ok")
    dist.destroy_process_group()


def test_engine_fleet_three_ranks_gloo(tmp_path):
    world, port = 3, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "front") and all(os.path.exists(tmp_path / f"worker{r}") for r in (1, 2))
