"""Continuous batching at decode-step granularity (the reference has none: its server runs one
`generate` per HTTP request, server.py:17).

One GPU, one captured decode graph over a fixed number of SLOTS (batch rows).  Every row has its own left padding and
its own position table, so rows are independent sequences that merely share the cache column they write next:

  * a waiting request is PREFILLED INTO A FREE ROW between two decode steps, right-aligned to the engine's current column
    (`model.prefill_slot`): left padding = column - prompt length, positions 0..S-1 from there on -- exactly the
    geometry `_tokenize` gives a left-padded batch row (phi.py:238-240), so the row computes what a B = 1 `generate` of
    the request computes (pad keys get zero weight, Q7);
  * every step is ONE graph replay for all rows; finished rows (EOS or budget, phi_3_vision_mlx.py:105-117, :390) are
    released at once and their slot is reusable at the next step -- nobody waits for the slowest row of a batch;
  * a request joins only if prompt <= column and column + max_tokens <= window; when the engine is idle the column
    jumps to the longest waiting prompt (an empty engine has no column to respect).

RoPE regime (phi.py:492: ONE short/long factor choice per call, from prompt + max_tokens > 4096).  An engine instance
serves ONE regime: `window <= 4096` -> short factors and only requests with S + max_tokens <= 4096 (what each of them
alone would pick); `window > 4096` -> long factors and only requests with S + max_tokens > 4096.  `RegimeRouter` puts
one engine of each kind behind a single `submit` and steps both from one thread.  With `quantize_cache=True` models the
slot state keeps the int8 KV cache of BASELINE config 5 (`model.new_slot_state`).

Admission is FIFO with bounded overtaking: a request that does not fit the current column is skipped by newer ones for
at most `patience` decode steps; after that nothing newer is admitted, the engine drains, and the column moves to the
blocked request.  A prefill group that fails is retried one request at a time, so one bad request fails alone.

Failures are loud: an exception inside a step fails every active and waiting request at once (nobody waits for a
timeout), the slot state and its captured graph are rebuilt (the in-launch split-KV merge relies on an all-sentinel
workspace between launches; a failed step may have left it dirty), and if THAT fails the engine marks itself dead and
refuses new work.  A negative token id (the device's report of NaN logits or a timed-out in-launch merge,
include/p3v.h) on an active row fails that request and re-arms the workspace.  A request whose waiter gave up
(`cancel()`) leaves its slot at the next step.

Cost model: a row reads the cache columns [0, column) whatever its own length (static split ranges), so a short request
that joins late pays for the padding it attends over with zero weight; the weights (7.4 GB/step) are shared by all rows.
"""
import collections
import threading

import numpy as np
import torch

ID_EOS = 32007
ROPE_WINDOW = 4096            # original_max_position_embeddings: the short / long factor boundary (phi.py:492)


def _is_gpu(device):
    return torch.device(device).type == "cuda"


def _sync(device):
    if _is_gpu(device):
        torch.cuda.synchronize(device)


def _set_device(device):
    if _is_gpu(device):                                         # a new thread starts on GPU 0 whatever the loader used
        torch.cuda.set_device(device)


class Request:
    __slots__ = ("inputs", "max_tokens", "tokens", "done", "row", "error", "S", "cancelled", "blocked_at")

    def __init__(self, inputs, max_tokens):
        self.inputs, self.max_tokens = inputs, int(max_tokens)
        self.S = int(np.asarray(inputs["input_ids"]).shape[-1])
        self.tokens, self.row, self.error = [], None, None
        self.cancelled, self.blocked_at = False, None
        self.done = threading.Event()

    def cancel(self):
        """The waiter gave up (timeout, client gone): the engine drops the request at its next step."""
        self.cancelled = True

    def fail(self, error):
        self.error = error
        self.done.set()


class ContinuousEngine:
    def __init__(self, model, processor, slots=8, window=ROPE_WINDOW, patience=64):
        self.model, self.processor, self.slots, self.window, self.patience = model, processor, slots, window, patience
        self.long_rope = window > ROPE_WINDOW
        self.rows = [None] * slots                               # row -> active Request
        self.waiting = collections.deque()
        self.lock = threading.Lock()
        self.steps = 0                                           # decode steps replayed (observability / tests)
        self.joined_mid_flight = 0                               # requests admitted while other rows were generating
        self.failures = 0                                        # steps that raised (each one rebuilt the slot state)
        self.dead = None                                         # the exception that made the engine unusable, if any
        self._new_state()

    def _new_state(self):
        self.st = self.model.new_slot_state(self.slots, self.window)
        self.model.serving = True                               # (the owner is a server: model.py)
        self.st.serving = True                                  # long-lived: the split-KV merge must not lean on dispatch order (model._split_plan)
        self.cache = [type("L", (), {"state": self.st})()]      # greedy_step reads cache[0].state

    # ---- request side (any thread)
    def accepts(self, S, max_tokens):
        """Does a request of this shape belong to this engine's RoPE regime and fit its window?"""
        if max_tokens < 1 or S < 1 or S + max_tokens > self.window:
            return False
        return (S + max_tokens > ROPE_WINDOW) == self.long_rope

    def submit(self, inputs, max_tokens):
        """inputs: a B = 1 `processor(text[, images])` result.  Returns the Request; wait on `.done`, read `.tokens`."""
        r = Request(inputs, max_tokens)
        if self.dead is not None:
            r.fail(RuntimeError(f"engine is down: {self.dead!r}"))
        elif not self.accepts(r.S, r.max_tokens):
            r.fail(ValueError(f"prompt {r.S} + max_tokens {r.max_tokens} is outside this engine "
                              f"({'long' if self.long_rope else 'short'}-RoPE regime, window {self.window})"))
        else:
            with self.lock:
                self.waiting.append(r)
        return r

    # ---- engine side (ONE thread)
    def _active(self):
        return [r for r in self.rows if r is not None]

    def _release(self, r):
        if r.row is not None and self.rows[r.row] is r:
            self.rows[r.row] = None
            self.st.pad_len[r.row:r.row + 1].fill_(self.window)  # every key masked: the row idles at zero cost of correctness
        r.done.set()

    _finish = _release

    def _fail_all(self, error):
        with self.lock:
            waiting, self.waiting = list(self.waiting), collections.deque()
        for r in self._active() + waiting:
            r.error = error
            if r.row is not None and self.rows[r.row] is r:
                self.rows[r.row] = None
            r.done.set()

    def _recover(self, error):
        """A step raised: fail everybody now, then rebuild the slot state + graph (or die loudly)."""
        self.failures += 1
        self._fail_all(error)
        try:
            _sync(self.model.device)
        except Exception:                                       # noqa: BLE001 -- a sticky HIP error surfaces again below
            pass
        try:
            self.st.graphs.clear()
            self.st = self.cache = None                         # free the old cache before the new one is allocated
            self._new_state()
        except Exception as e:                                  # noqa: BLE001
            self.dead = e
            self._fail_all(RuntimeError(f"engine is down: {e!r}"))

    def _rearm_workspace(self):
        """The in-launch split-KV merge expects an all-sentinel workspace (ops.attention_ws); after a poisoned step a late
        partial may have landed on top of the restored sentinels -- refill before the next replay."""
        g = self.st.graphs.get("greedy")
        ws = g and g["bufs"].get("ws")
        if ws is not None:
            _sync(self.model.device)
            ws.view(torch.int32).fill_(-1)

    def _pick(self):
        """FIFO admission with bounded overtaking (under the lock).  Returns (requests to prefill, their free rows)."""
        st = self.st
        for r in self.waiting:
            if r.cancelled:
                r.done.set()
        self.waiting = collections.deque(r for r in self.waiting if not r.cancelled)
        if not self.waiting:
            return [], []
        if not self._active():
            # idle: the column follows the OLDEST waiting prompts -- but never so far right that the HEAD request no longer
            # fits (column + its max_tokens <= window).  The head itself always qualifies (`accepts`), so an idle engine
            # admits at least the head: a blocked / draining head cannot keep everybody out for ever.
            first = list(self.waiting)[:self.slots]
            head = first[0]
            st.offset = max(r.S for r in first if r.S == head.S or r.S + head.max_tokens <= self.window)
        free = [i for i, r in enumerate(self.rows) if r is None]
        admit, keep, draining = [], collections.deque(), False
        for r in self.waiting:
            fits = r.S <= st.offset and st.offset + r.max_tokens <= self.window
            if not draining and fits and len(admit) < len(free):
                admit.append(r)
                continue
            if not fits and not draining:
                if r.blocked_at is None:
                    r.blocked_at = self.steps
                draining = self.steps - r.blocked_at >= self.patience       # nobody newer gets in: the engine drains for r
            keep.append(r)
        self.waiting = keep
        return admit, free

    def _prefill_group(self, group, row0, busy):
        from .processor import collate_requests
        st, n = self.st, len(group)
        g = self.model.decode_graph(st)
        for i, r in enumerate(group):
            r.row = row0 + i
        toks = self.model.prefill_slot(st, row0, collate_requests([r.inputs for r in group]) if n > 1 else group[0].inputs)
        first = toks.reshape(-1).tolist()
        if min(first) < 0:
            raise RuntimeError(f"device prefill failed: NaN logits (token ids {first})")
        g["tok"][row0:row0 + n].copy_(toks.reshape(-1))
        for r, t in zip(group, first):
            self.rows[r.row] = r
            self.joined_mid_flight += int(busy)
            r.tokens.append(t)
            if t == ID_EOS or len(r.tokens) >= r.max_tokens:
                self._release(r)

    def _admit(self):
        with self.lock:
            admit, free = self._pick()
        if not admit:
            return
        busy = bool(self._active())
        # requests of nearly equal length that get ADJACENT free rows are prefilled as one left-padded group (one pass over
        # the weights instead of one per request; dist.GROUP_PAD bounds the padding a request may carry)
        from .dist import GROUP_PAD
        admit.sort(key=lambda r: -r.S)
        free.sort()
        while admit:
            run = 1
            while run < len(free) and free[run] == free[0] + run:
                run += 1
            n = 1
            while n < min(run, len(admit)) and admit[0].S - admit[n].S <= GROUP_PAD:
                n += 1
            group, admit = admit[:n], admit[n:]
            row0, free = free[0], free[n:]
            try:
                self._prefill_group(group, row0, busy)
            except Exception as e:                              # noqa: BLE001 -- reported to the request(s), the engine lives on
                for r in group:
                    if self.rows[r.row] is r:
                        self.rows[r.row] = None
                    self.st.pad_len[r.row:r.row + 1].fill_(self.window)
                if n == 1:
                    group[0].fail(e)
                    continue
                for i, r in enumerate(group):                   # one bad request must not fail its neighbours: retry alone
                    try:
                        self._prefill_group([r], row0 + i, busy)
                    except Exception as e1:                     # noqa: BLE001
                        self.st.pad_len[r.row:r.row + 1].fill_(self.window)
                        r.fail(e1)

    def step(self):
        """Admit what fits, then one decode step for every active row.  Returns the number of active rows."""
        if self.dead is not None:
            return 0
        self._admit()
        for r in self._active():
            if r.cancelled:
                self._release(r)
        active = self._active()
        if not active:
            return 0
        g = self.model.decode_graph(self.st)
        _, tok = self.model.greedy_step(g["host_tok"] if g["host_tok"] is not None else g["tok"].view(-1, 1), self.cache)
        rows = tok.reshape(-1).tolist()                          # ONE D2H copy per step (the reference's mx.eval)
        self.steps += 1
        poisoned = False
        for r in active:
            t = rows[r.row]
            if t < 0:                                            # the device's report of a failed step for this row
                poisoned = True
                r.error = RuntimeError(f"device step failed: NaN logits or split merge timeout (token id {t})")
                self._release(r)
                continue
            r.tokens.append(t)
            if t == ID_EOS or len(r.tokens) >= r.max_tokens:
                self._release(r)
        if poisoned:
            self._rearm_workspace()
        if self.st.offset + 1 > self.st.T:                       # window exhausted: budgets were checked at admission,
            for r in self._active():                             # so nothing can still be running -- belt and braces
                self._release(r)
        return len(active)

    def safe_step(self):
        """`step` that never raises: an exception fails all requests at once and rebuilds the state (see module doc)."""
        try:
            return self.step()
        except Exception as e:                                  # noqa: BLE001
            self._recover(e)
            return 0

    def run_until_idle(self, max_steps=1 << 20):
        n = 0
        while n < max_steps and (self.step() or self.waiting):
            n += 1
        return n

    def serve_forever(self, stop_event, idle_sleep=0.002):
        """Engine thread body: step while there is work, nap when idle."""
        _set_device(self.model.device)
        while not stop_event.is_set():
            if not self.safe_step() and not self.waiting:
                stop_event.wait(idle_sleep)

    # ---- convenience: text in, text out (what the HTTP handler calls)
    def generate(self, prompts, images=None, max_tokens=512, timeout=600.0):
        return _generate_text(self, self.processor, prompts, images, max_tokens, timeout)


def _generate_text(engine, processor, prompts, images, max_tokens, timeout):
    from . import api
    prompts = [prompts] if isinstance(prompts, str) else list(prompts)
    images = images if images is not None else [None] * len(prompts)
    reqs = []
    for p, im in zip(prompts, images):
        text, imgs = api._apply_chat_template(p, im, False)
        reqs.append(engine.submit(processor(text, imgs) if imgs is not None else processor(text), max_tokens))
    out = []
    try:
        for r in reqs:
            if not r.done.wait(timeout):
                raise TimeoutError("engine did not finish the request in time")
            if r.error is not None:
                raise r.error
            ids = r.tokens[:r.tokens.index(ID_EOS) + 1] if ID_EOS in r.tokens else r.tokens
            out.append(processor.tokenizer.decode(ids))
    except BaseException:
        for r in reqs:                                          # nobody is waiting for these any more: free their slots
            if not r.done.is_set():
                r.cancel()
        raise
    return out


class RegimeRouter:
    """One `submit` in front of a short-RoPE engine (window 4096) and a long-RoPE engine (window > 4096), both stepped by
    one thread: a request goes to the engine whose regime it would pick on its own (phi.py:492), so its tokens are what a
    solo `generate` of it produces."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.processor = self.engines[0].processor

    def submit(self, inputs, max_tokens):
        S = int(np.asarray(inputs["input_ids"]).shape[-1])
        for e in self.engines:
            if e.accepts(S, int(max_tokens)):
                return e.submit(inputs, max_tokens)
        r = Request(inputs, max_tokens)
        r.fail(ValueError(f"prompt {S} + max_tokens {max_tokens} fits no engine window"))
        return r

    @property
    def waiting(self):
        return any(e.waiting for e in self.engines)

    def safe_step(self):
        return sum(e.safe_step() for e in self.engines)

    def serve_forever(self, stop_event, idle_sleep=0.002):
        _set_device(self.engines[0].model.device)
        while not stop_event.is_set():
            if not self.safe_step() and not self.waiting:
                stop_event.wait(idle_sleep)

    def generate(self, prompts, images=None, max_tokens=512, timeout=600.0):
        return _generate_text(self, self.processor, prompts, images, max_tokens, timeout)
