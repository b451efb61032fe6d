"""Host logic of the continuous-batching engine (engine.py) on a CPU stand-in for the model's slot interface:
admission order, regime routing, failure handling (a step that raises must fail everybody at once and the
engine must stay usable; negative token ids; cancelled waiters).  The token arithmetic of the real model is pinned by
the GPU tests against the oracle fixtures (tests/test_model_gpu.py)."""
import threading
import time

import numpy as np
import pytest
import torch

from phi_3_vision_mlx_amd.engine import ContinuousEngine, RegimeRouter

EOS = 32007


class _State:
    def __init__(self, slots, window):
        self.pad_len = torch.full((slots,), window, dtype=torch.int32)
        self.offset, self.T, self.graphs = 0, window, {}


class SlotStub:
    """Each row's tokens are a function of its own prompt and step only (like the real model: rows are independent)."""
    device = "cpu"

    def __init__(self):
        self.states, self.fail_next_step, self.poison_row, self.fail_new_state = 0, False, None, False
        self.prefills = []                                   # (row, [prompt lengths]) per prefill_slot call

    def new_slot_state(self, slots, window):
        if self.fail_new_state:
            raise RuntimeError("out of memory")
        self.states += 1
        st = _State(slots, window)
        st.key, st.step = np.zeros(slots, dtype=np.int64), np.zeros(slots, dtype=np.int64)
        return st

    def decode_graph(self, st):
        g = st.graphs.get("greedy")
        if g is None:
            g = st.graphs["greedy"] = {"tok": torch.zeros(len(st.pad_len), dtype=torch.int32), "host_tok": None,
                                       "bufs": {"ws": torch.full((8,), -1, dtype=torch.int32).view(torch.float32)}}
        return g

    def _tok(self, st):
        t = (st.key * 31 + st.step * 7919) % 31000 + 3
        return np.where((st.key + st.step) % 29 == 28, EOS, t)

    def prefill_slot(self, st, row, inputs):
        ids = np.asarray(inputs["input_ids"])
        ids = ids[None] if ids.ndim == 1 else ids
        n, S = ids.shape
        m = np.asarray(inputs["mask"]).reshape(n, S) if "mask" in inputs else np.ones_like(ids)
        if (ids == 666).any():
            raise ValueError("bad request")
        assert S <= st.offset
        self.prefills.append((row, m.sum(1).tolist()))
        st.key[row:row + n] = (ids * m).sum(1)
        st.step[row:row + n] = 0
        st.pad_len[row:row + n] = torch.as_tensor(st.offset - m.sum(1), dtype=torch.int32)
        return torch.as_tensor(self._tok(st)[row:row + n, None].astype(np.int32))

    def greedy_step(self, token, cache):
        st = cache[0].state
        if self.fail_next_step:
            self.fail_next_step = False
            raise RuntimeError("HIP error: illegal memory access")
        st.step += 1
        st.offset += 1
        t = self._tok(st).astype(np.int32)
        if self.poison_row is not None:
            t[self.poison_row] = -1
            self.poison_row = None
        g = self.decode_graph(st)
        g["host_tok"] = torch.as_tensor(t[:, None])
        return None, g["host_tok"]


def req(n, seed=0, first=5):
    ids = np.random.default_rng(seed).integers(3, 600, (1, n)).astype(np.int64)
    ids[0, 0] = first
    return {"input_ids": ids}


def solo(inputs, max_tokens):
    """What one request alone produces: the stub's token rule on its own key."""
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=1, window=4096)
    r = e.submit(inputs, max_tokens)
    e.run_until_idle()
    return r.tokens


def test_rows_are_independent_and_leave_at_their_own_eos():
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=3, window=4096)
    reqs = [(req(40, 1), 30), (req(17, 2), 9), (req(40, 3), 25), (req(33, 4), 12), (req(8, 5), 30)]
    rs = [e.submit(*reqs[0]), e.submit(*reqs[1])]
    e.step(), e.step()
    rs += [e.submit(*q) for q in reqs[2:]]                   # join while others are mid-flight
    e.run_until_idle()
    for r, (inp, mt) in zip(rs, reqs):
        assert r.done.is_set() and r.error is None
        assert r.tokens == solo(inp, mt)
        assert len(r.tokens) == mt or r.tokens[-1] == EOS
    assert e.joined_mid_flight >= 2
    assert any(len(lens) > 1 for _, lens in m.prefills)      # near-equal lengths shared one prefill pass


def test_fifo_with_bounded_overtaking():
    """A long request behind short traffic is overtaken for at most `patience` steps, then the engine drains for it."""
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=2, window=4096, patience=5)
    first = e.submit(req(20, 1), 400)
    e.step()                                                 # column = 20
    long_ = e.submit(req(500, 2), 5)                         # does not fit left of column ~20
    shorts, n_before = [], None
    for i in range(60):
        shorts.append(e.submit(req(10, 10 + i), 2))
        e.step()
        if long_.tokens and n_before is None:
            n_before = sum(1 for s in shorts if s.tokens)
    e.run_until_idle()
    assert long_.done.is_set() and long_.error is None and long_.tokens == solo(req(500, 2), 5)
    # it was overtaken by a few shorts (patience), never by all of them: it ran while newer shorts were still waiting
    assert 1 <= n_before < 20, n_before
    assert all(s.done.is_set() and s.error is None for s in shorts)
    assert first.tokens == solo(req(20, 1), 400)[:len(first.tokens)]


def test_step_exception_fails_everyone_now_and_engine_recovers():
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=2, window=4096)
    a, b, c = e.submit(req(20, 1), 50), e.submit(req(20, 2), 50), e.submit(req(20, 3), 50)
    e.step()
    live = [r for r in (a, b, c) if not r.done.is_set()]
    assert len(live) >= 2 and c in live
    m.fail_next_step = True
    assert e.safe_step() == 0
    for r in live:                                           # active AND waiting requests: failed at once, no timeout
        assert r.done.is_set() and "illegal memory access" in str(r.error)
    assert e.failures == 1 and m.states == 2 and e.dead is None      # the slot state was rebuilt
    d = e.submit(req(20, 4), 10)
    e.run_until_idle()
    assert d.error is None and d.tokens == solo(req(20, 4), 10)
    # a rebuild that fails: the engine is dead, says so, and refuses new work immediately
    m.fail_next_step, m.fail_new_state = True, True
    f = e.submit(req(20, 5), 10)
    e.safe_step(), e.safe_step()
    assert f.done.is_set() and f.error is not None and e.dead is not None
    g = e.submit(req(20, 6), 10)
    assert g.done.is_set() and "engine is down" in str(g.error)
    assert e.safe_step() == 0


def test_negative_token_fails_that_row_only_and_rearms_the_workspace():
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=2, window=4096)
    a, b = e.submit(req(20, 1), 30), e.submit(req(20, 2), 30)
    e.step()
    e.st.graphs["greedy"]["bufs"]["ws"].view(torch.int32)[3] = 12345    # a stale partial a late split left behind
    m.poison_row = a.row
    e.step()
    assert a.done.is_set() and a.error is not None and -1 not in a.tokens
    assert (e.st.graphs["greedy"]["bufs"]["ws"].view(torch.int32) == -1).all()
    e.run_until_idle()
    assert b.error is None and b.tokens == solo(req(20, 2), 30)


def test_bad_request_in_a_group_fails_alone():
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=4, window=4096)
    good1, bad, good2 = req(30, 1), req(30, 2), req(29, 3)
    bad["input_ids"][0, 5] = 666
    rs = [e.submit(good1, 8), e.submit(bad, 8), e.submit(good2, 8)]
    e.run_until_idle()
    assert isinstance(rs[1].error, ValueError)
    assert rs[0].error is None and rs[0].tokens == solo(good1, 8)
    assert rs[2].error is None and rs[2].tokens == solo(good2, 8)


def test_cancelled_requests_leave_the_queue_and_their_slot():
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=1, window=4096)
    a = e.submit(req(20, 1), 1000)
    w = e.submit(req(20, 2), 5)
    e.step(), e.step()
    w.cancel()
    e.step()
    assert w.done.is_set() and not w.tokens and not e.waiting
    a.cancel()
    e.step()
    assert a.done.is_set() and e.rows == [None]
    n = e.submit(req(20, 3), 4)
    e.run_until_idle()
    assert n.tokens == solo(req(20, 3), 4)


def test_generate_timeout_cancels_and_frees_slots():
    class Tok:
        def decode(self, ids):
            return " ".join(map(str, ids))

    class Proc:
        tokenizer = Tok()

        def __call__(self, text, images=None):
            return req(10 + len(text) % 7, len(text))

    m = SlotStub()
    e = ContinuousEngine(m, Proc(), slots=2, window=4096)
    with pytest.raises(TimeoutError):                        # nobody steps the engine: the waiter gives up ...
        e.generate(["a", "bb"], max_tokens=5, timeout=0.05)
    e.run_until_idle()                                       # ... and its requests are dropped, not run
    assert not e.waiting and e.rows == [None, None] and e.steps == 0
    stop = threading.Event()
    t = threading.Thread(target=e.serve_forever, args=(stop,), daemon=True)
    t.start()
    out = e.generate(["a", "bb"], max_tokens=5, timeout=10)
    stop.set(), t.join(2)
    assert len(out) == 2 and all(isinstance(s, str) and s for s in out)


def test_regime_router_sends_each_request_to_its_own_rope_regime():
    m = SlotStub()
    short, long_ = ContinuousEngine(m, None, slots=2, window=4096), ContinuousEngine(m, None, slots=2, window=8192)
    assert short.accepts(100, 3996) and not short.accepts(100, 3997) and not short.accepts(5000, 10)
    assert long_.accepts(100, 3997) and long_.accepts(5000, 100) and not long_.accepts(100, 100) and not long_.accepts(8000, 200)
    router = RegimeRouter([short, long_])
    a, b = router.submit(req(100, 1), 50), router.submit(req(4090, 2), 50)
    c = router.submit(req(9000, 3), 50)
    assert c.done.is_set() and isinstance(c.error, ValueError)
    while router.safe_step() or router.waiting:
        pass
    assert a.tokens == solo(req(100, 1), 50) and short.steps > 0
    assert b.error is None and len(b.tokens) > 0 and long_.steps > 0
    direct = short.submit(req(4090, 2), 50)                 # the wrong engine refuses instead of silently using short factors
    assert isinstance(direct.error, ValueError)


def test_idle_engine_always_admits_the_head_request():
    """a head request that needs nearly the whole window (small prompt, huge max_tokens) next to a
    longer prompt.  Once the engine is idle the column must be set so that the HEAD fits -- before the fix it followed the
    longest of the oldest waiters, the head stayed blocked, draining kept everybody else out and `steps` stopped."""
    m = SlotStub()
    e = ContinuousEngine(m, None, slots=2, window=4096, patience=5)
    a = e.submit(req(10, 1), 4000)                           # fits only with the column at <= 96
    c = e.submit(req(500, 2), 12)
    for _ in range(8):
        e.step()
    late = e.submit(req(400, 3), 5)
    for _ in range(6000):
        if a.done.is_set() and c.done.is_set() and late.done.is_set():
            break
        e.step()
    for r, (inp, mt) in ((a, (req(10, 1), 4000)), (c, (req(500, 2), 12)), (late, (req(400, 3), 5))):
        assert r.done.is_set() and r.error is None
        assert r.tokens == solo(inp, mt)
