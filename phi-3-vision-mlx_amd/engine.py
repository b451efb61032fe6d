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
    jumps to the longest waiting prompt (an empty engine has no column to respect); requests that need more than the
    window (4096 = the short-RoPE regime, phi.py:492) are not for this engine -- run them through `generate`.

Cost model: a row reads the cache columns [0, column) whatever its own length (static split ranges), so a short request
that joins late pays for the padding it attends over with zero weight; the weights (7.4 GB/step) are shared by all rows.
"""
import collections
import threading

import numpy as np
import torch

ID_EOS = 32007


class Request:
    __slots__ = ("inputs", "max_tokens", "tokens", "done", "row", "error", "S")

    def __init__(self, inputs, max_tokens):
        self.inputs, self.max_tokens = inputs, int(max_tokens)
        self.S = int(np.asarray(inputs["input_ids"]).shape[-1])
        self.tokens, self.row, self.error = [], None, None
        self.done = threading.Event()


class ContinuousEngine:
    def __init__(self, model, processor, slots=8, window=4096):
        self.model, self.processor, self.slots, self.window = model, processor, slots, window
        self.st = model.new_slot_state(slots, window)
        self.cache = [type("L", (), {"state": self.st})()]      # greedy_step reads cache[0].state
        self.rows = [None] * slots                               # row -> active Request
        self.waiting = collections.deque()
        self.lock = threading.Lock()
        self.steps = 0                                           # decode steps replayed (observability / tests)
        self.joined_mid_flight = 0                               # requests admitted while other rows were generating

    # ---- request side (any thread)
    def submit(self, inputs, max_tokens):
        """inputs: a B = 1 `processor(text[, images])` result.  Returns the Request; wait on `.done`, read `.tokens`."""
        r = Request(inputs, max_tokens)
        if r.S + r.max_tokens > self.window or r.max_tokens < 1:
            r.error = ValueError(f"prompt {r.S} + max_tokens {r.max_tokens} exceeds the engine window {self.window}")
            r.done.set()
            return r
        with self.lock:
            self.waiting.append(r)
        return r

    # ---- engine side (ONE thread)
    def _active(self):
        return [r for r in self.rows if r is not None]

    def _finish(self, r):
        self.rows[r.row] = None
        self.st.pad_len[r.row:r.row + 1].fill_(self.window)     # every key masked: the row idles at zero cost of correctness
        r.done.set()

    def _admit(self):
        st = self.st
        with self.lock:
            if not self.waiting:
                return
            if not self._active():                              # idle: the column follows the waiting prompts
                first = list(self.waiting)[:self.slots]
                st.offset = max(r.S for r in first)
            free = [i for i, r in enumerate(self.rows) if r is None]
            admit, keep = [], collections.deque()
            for r in self.waiting:
                if free and len(admit) < len(free) and r.S <= st.offset and st.offset + r.max_tokens <= self.window:
                    admit.append(r)
                else:
                    keep.append(r)
            self.waiting = keep
        g = self.model.decode_graph(st)
        busy = bool(self._active())
        # requests of nearly equal length that get ADJACENT free rows are prefilled as one left-padded group (one pass over
        # the weights instead of one per request; dist.GROUP_PAD bounds the padding a request may carry)
        from .dist import GROUP_PAD
        from .processor import collate_requests
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
            for i, r in enumerate(group):
                r.row = row0 + i
            try:
                toks = self.model.prefill_slot(st, row0, collate_requests([r.inputs for r in group]) if n > 1 else group[0].inputs)
            except Exception as e:                              # noqa: BLE001 -- reported to the requests, the engine lives on
                for r in group:
                    r.error = e
                    self.st.pad_len[r.row:r.row + 1].fill_(self.window)
                    r.done.set()
                continue
            g["tok"][row0:row0 + n].copy_(toks.reshape(-1))
            first = toks.reshape(-1).tolist()
            for r, t in zip(group, first):
                self.rows[r.row] = r
                self.joined_mid_flight += int(busy)
                r.tokens.append(t)
                if t == ID_EOS or len(r.tokens) >= r.max_tokens:
                    self._finish(r)

    def step(self):
        """Admit what fits, then one decode step for every active row.  Returns the number of active rows."""
        self._admit()
        active = self._active()
        if not active:
            return 0
        g = self.model.decode_graph(self.st)
        _, tok = self.model.greedy_step(g["host_tok"] if g["host_tok"] is not None else g["tok"].view(-1, 1), self.cache)
        rows = tok.reshape(-1).tolist()                          # ONE D2H copy per step (the reference's mx.eval)
        self.steps += 1
        for r in active:
            t = rows[r.row]
            r.tokens.append(t)
            if t == ID_EOS or len(r.tokens) >= r.max_tokens:
                self._finish(r)
        if self.st.offset + 1 > self.st.T:                       # window exhausted: budgets were checked at admission,
            for r in self._active():                             # so nothing can still be running -- belt and braces
                self._finish(r)
        return len(active)

    def run_until_idle(self, max_steps=1 << 20):
        n = 0
        while n < max_steps and (self.step() or self.waiting):
            n += 1
        return n

    def serve_forever(self, stop_event, idle_sleep=0.002):
        """Engine thread body: step while there is work, nap when idle."""
        torch.cuda.set_device(self.model.device)
        while not stop_event.is_set():
            if not self.step() and not self.waiting:
                stop_event.wait(idle_sleep)

    # ---- convenience: text in, text out (what the HTTP handler calls)
    def generate(self, prompts, images=None, max_tokens=512, timeout=600.0):
        from . import api
        prompts = [prompts] if isinstance(prompts, str) else list(prompts)
        images = images if images is not None else [None] * len(prompts)
        reqs = []
        for p, im in zip(prompts, images):
            text, imgs = api._apply_chat_template(p, im, False)
            reqs.append(self.submit(self.processor(text, imgs) if imgs is not None else self.processor(text), max_tokens))
        out = []
        for r in reqs:
            if not r.done.wait(timeout):
                raise TimeoutError("engine did not finish the request in time")
            if r.error is not None:
                raise r.error
            ids = r.tokens[:r.tokens.index(ID_EOS) + 1] if ID_EOS in r.tokens else r.tokens
            out.append(self.processor.tokenizer.decode(ids))
        return out
