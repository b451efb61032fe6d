"""Continuous-batching engines composed with the batch-sharded deployment: one process per GPU, weights replicated
(BASELINE.json north_star), every rank stepping its OWN engine.ContinuousEngine (or RegimeRouter); rank 0 owns the request
surface (the HTTP handler's `submit`) and hands each request to the rank with the fewest requests outstanding.  This is
SURVEY.md 8 row f1's "request queue feeding the batch-sharded engine" for the continuous engine (the one-shot path is
dist.generate_sharded); the reference has no counterpart (its server.py serves one process).

No data-path collective: a request is independent of every other one, so the only traffic between ranks is the request itself
(token ids, and the pre-processed crops for an image request) and its token list coming back -- host memory over two gloo
groups, one per direction (a send is a size message plus a payload message: one lock per destination keeps the pairs together,
one receiver thread per source takes them in order).  RCCL is not involved; the GPUs never talk.

    groups = fleet.make_groups()                      # every rank, after torch.distributed.init_process_group(...)
    if rank: fleet.worker(engine, groups)             # blocks until rank 0 closes the fleet
    else:    front = fleet.EngineFleet(engine, groups, world)   # same surface as an engine: submit / generate / serve_forever
"""
import threading
import time

from .engine import Request, _generate_text

_ERRORS = {"ValueError": ValueError, "TimeoutError": TimeoutError, "TypeError": TypeError}   # what the HTTP handler tells apart


IDLE_TIMEOUT_DAYS = 3650      # a server may sit idle for any length of time: a blocking recv must not expire (gloo's default: 30 min)


def make_groups(timeout=None):
    """(rank 0 -> workers, workers -> rank 0): two gloo groups over all ranks.  Collective: every rank calls it once.
    The groups carry blocking point-to-point receives that wait for the NEXT REQUEST, so their timeout is "never": gloo tears
    the pair down when a receive times out (the next send fails with "Connection closed by peer" -- measured, so a timeout
    cannot be caught and retried), while a peer that really dies still surfaces at once as a closed connection."""
    import datetime

    import torch.distributed as dist
    timeout = timeout or datetime.timedelta(days=IDLE_TIMEOUT_DAYS)
    return dist.new_group(backend="gloo", timeout=timeout), dist.new_group(backend="gloo", timeout=timeout)


def _send(obj, dst, group):
    import torch.distributed as dist
    dist.send_object_list([obj], dst=dst, group=group)


def _recv(src, group):
    import torch.distributed as dist
    box = [None]
    dist.recv_object_list(box, src=src, group=group)
    return box[0]


class RemoteRequest(Request):
    """The rank-0 handle of a request that runs on another rank: same fields as engine.Request, filled when the result arrives."""
    __slots__ = ("fleet", "rank", "rid")

    def cancel(self):
        Request.cancel(self)
        try:
            self.fleet._send_to(self.rank, ("cancel", self.rid))
        except Exception:                                        # noqa: BLE001  the peer is gone: its receiver fails the request
            pass


class EngineFleet:
    """Rank 0's front: `submit` picks the least-loaded rank (its own engine included; ties go to the lowest rank)."""

    def __init__(self, engine, groups, world):
        self.engine, self.processor, self.world = engine, engine.processor, int(world)
        self.down, self.up = groups
        self.lock = threading.Lock()
        self.send_locks = {r: threading.Lock() for r in range(1, self.world)}
        self.pending, self.local, self.next_id = {}, [], 0      # rid -> RemoteRequest; live local handles
        self.load = [0] * self.world                            # requests outstanding per remote rank (index 0 unused)
        self.sent = [0] * self.world                            # requests ever handed to each rank (observability / tests)
        self.closed = False
        self.receivers = [threading.Thread(target=self._receive, args=(r,), daemon=True) for r in range(1, self.world)]
        for t in self.receivers:
            t.start()

    # ---- request side (any thread)
    def _send_to(self, rank, msg, force=False):
        with self.send_locks[rank]:
            if force or not self.closed:
                _send(msg, rank, self.down)

    def submit(self, inputs, max_tokens):
        with self.lock:
            self.local = [h for h in self.local if not h.done.is_set()]
            loads = [len(self.local)] + self.load[1:]
            dst = min(range(self.world), key=lambda r: (loads[r], r))
            self.sent[dst] += 1
            if dst == 0:
                h = self.engine.submit(inputs, max_tokens)
                self.local.append(h)
                return h
            h = RemoteRequest(inputs, max_tokens)
            h.fleet, h.rank, h.rid = self, dst, self.next_id
            self.next_id += 1
            self.pending[h.rid] = h
            self.load[dst] += 1
        try:
            self._send_to(dst, ("submit", h.rid, inputs, int(max_tokens)))
        except Exception as e:                                   # noqa: BLE001  the request never left: fail it here
            with self.lock:
                self.pending.pop(h.rid, None)
                self.load[dst] = 1 << 30
            h.fail(RuntimeError(f"rank {dst} is unreachable ({type(e).__name__}: {e})"))
        return h

    def _receive(self, rank):
        why = "stopped"
        while True:
            try:
                msg = _recv(rank, self.up)
            except Exception as e:                               # noqa: BLE001  the peer died / the group was torn down
                why = f"is unreachable ({type(e).__name__}: {e})"
                break
            if msg[0] == "bye":
                break
            _, rid, tokens, err = msg
            with self.lock:
                h = self.pending.pop(rid, None)
                self.load[rank] -= 1
            if h is None:
                continue
            h.tokens = list(tokens)
            if err is None:
                h.done.set()
            else:
                h.fail(_ERRORS.get(err[0], RuntimeError)(f"rank {rank}: {err[1]}"))
        with self.lock:                                          # the worker is gone: nobody will answer what it still held,
            self.load[rank] = 1 << 30                            # and nothing more goes there
            lost = [h for h in self.pending.values() if h.rank == rank]
            for h in lost:
                self.pending.pop(h.rid, None)
        for h in lost:
            h.fail(RuntimeError(f"rank {rank} {why} before the request finished"))

    # ---- the engine surface the HTTP backend drives (server.ContinuousBackend): rank 0 steps its own engine
    @property
    def waiting(self):
        return self.engine.waiting

    def safe_step(self):
        return self.engine.safe_step()

    def serve_forever(self, stop_event, idle_sleep=0.002):
        self.engine.serve_forever(stop_event, idle_sleep)

    def generate(self, prompts, images=None, max_tokens=512, timeout=600.0):
        return _generate_text(self, self.processor, prompts, images, max_tokens, timeout)

    def close(self, timeout=10.0):
        """Tell every worker to stop (each answers "bye" once its engine thread is down)."""
        self.closed = True                                       # first: nothing new goes out, a failing stop cannot leave us open
        for r in range(1, self.world):
            if self.load[r] >= 1 << 30:                          # already known dead
                continue
            try:
                self._send_to(r, ("stop",), force=True)
            except Exception:                                    # noqa: BLE001  a dead peer: its receiver thread has failed its requests
                pass
        for t in self.receivers:
            t.join(timeout)


def worker(engine, groups, poll_s=0.002):
    """Ranks > 0: step the local engine, take requests from rank 0, send each result back when its handle completes.
    Returns after rank 0's `close()`."""
    down, up = groups
    stop = threading.Event()
    stepper = threading.Thread(target=engine.serve_forever, args=(stop,), daemon=True)
    stepper.start()
    live, lock, closing = {}, threading.Lock(), threading.Event()

    def completions():
        while True:
            with lock:
                done = [(rid, h) for rid, h in live.items() if h.done.is_set()]
                for rid, _ in done:
                    del live[rid]
                idle = not live
            for rid, h in done:
                err = None if h.error is None else (type(h.error).__name__, str(h.error))
                _send(("done", rid, [int(t) for t in h.tokens], err), 0, up)
            if closing.is_set() and idle and not done:
                _send(("bye",), 0, up)                            # from THIS thread: the only sender on `up`, so the size / payload
                return                                           # pairs of two messages can never interleave
            if not done:
                time.sleep(poll_s)

    reporter = threading.Thread(target=completions, daemon=True)
    reporter.start()
    while True:
        msg = _recv(0, down)
        if msg[0] == "stop":
            break
        if msg[0] == "submit":
            h = engine.submit(msg[2], msg[3])
            with lock:
                live[msg[1]] = h
        elif msg[0] == "cancel":
            with lock:
                h = live.get(msg[1])
            if h is not None:
                h.cancel()
    stop.set()                                                   # the engine stops stepping: what is still live cannot finish
    stepper.join(10.0)
    with lock:
        left = list(live.values())
    for h in left:
        if not h.done.is_set():
            h.fail(RuntimeError("worker stopped"))
    closing.set()
    reporter.join()
