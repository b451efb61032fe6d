"""HTTP façade over `generate` -- the endpoint contract of the reference's server.py (server.py:8-29):

    POST /v1/completions   {"prompt": str | [str, ...], "max_tokens": int (default 512)}
      -> 200 {"model": "phi-3-vision", "responses": [str, ...]}            anything else -> 404

with one difference in the plumbing: requests do not call the model from the HTTP thread.  They go into a queue that a
single engine thread drains (the model object holds one in-flight sequence group), and requests that are
waiting at the same time and ask for the same `max_tokens` are merged into ONE batched `generate` call (the batch
dimension is what shards across GPUs, `dist.generate_sharded`).  Malformed bodies get 400 instead of a dropped connection.

    python -m phi_3_vision_mlx_amd.server --port 8000 [--synthetic] [--blind]
"""
import json
import queue
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

MODEL_NAME = "phi-3-vision"


class _Job:
    __slots__ = ("prompts", "max_tokens", "done", "result", "error")

    def __init__(self, prompts, max_tokens):
        self.prompts, self.max_tokens = prompts, max_tokens
        self.done, self.result, self.error = threading.Event(), None, None


class EngineQueue:
    """Single consumer in front of a non-re-entrant `generate_fn(prompts: list[str], max_tokens) -> str | list[str]`."""

    def __init__(self, generate_fn, max_batch=64, window_s=0.005):
        self.generate_fn, self.max_batch, self.window_s = generate_fn, max_batch, window_s
        self.jobs = queue.Queue()
        self.batches = []                       # sizes of the generate calls issued (observability / tests)
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, prompts, max_tokens):
        job = _Job(prompts, max_tokens)
        self.jobs.put(job)
        job.done.wait()
        if job.error is not None:
            raise job.error
        return job.result

    def close(self):
        self._stop = True
        self.jobs.put(None)
        self.thread.join(timeout=5)

    def _collect(self, first):
        """`first` plus every queued job with the same max_tokens that fits, waiting at most `window_s` for stragglers."""
        group, n, held = [first], len(first.prompts), []
        while n < self.max_batch:
            try:
                job = self.jobs.get(timeout=self.window_s)
            except queue.Empty:
                break
            if job is None:
                self.jobs.put(None)
                break
            if job.max_tokens == first.max_tokens and n + len(job.prompts) <= self.max_batch:
                group.append(job)
                n += len(job.prompts)
            else:
                held.append(job)
        for job in held:                        # different budget: next round, order kept
            self.jobs.put(job)
        return group

    def _run(self):
        while not self._stop:
            first = self.jobs.get()
            if first is None:
                break
            group = self._collect(first)
            flat = [p for j in group for p in j.prompts]
            try:
                out = self.generate_fn(flat, first.max_tokens)
                out = [out] if isinstance(out, str) else list(out)
                if len(out) != len(flat):
                    raise RuntimeError(f"generate returned {len(out)} texts for {len(flat)} prompts")
                self.batches.append(len(flat))
                i = 0
                for j in group:
                    j.result = out[i:i + len(j.prompts)]
                    i += len(j.prompts)
            except Exception as e:              # noqa: BLE001 -- reported to every waiting request
                for j in group:
                    j.error = e
            for j in group:
                j.done.set()


def make_handler(engine):
    class CompletionHandler(BaseHTTPRequestHandler):
        def _send(self, code, payload):
            body = json.dumps(payload).encode("utf-8")
            self.send_response(code)
            self.send_header("Content-Type", "application/json")
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def do_POST(self):
            if self.path != "/v1/completions":
                self.send_error(404, "Not Found")
                return
            try:
                request = json.loads(self.rfile.read(int(self.headers.get("Content-Length", 0))).decode("utf-8"))
                prompts = request.get("prompt", "")
                max_tokens = int(request.get("max_tokens", 512))
                prompts = [prompts] if isinstance(prompts, str) else list(prompts)
                if not prompts or not all(isinstance(p, str) for p in prompts):
                    raise ValueError("prompt must be a string or a list of strings")
            except (ValueError, TypeError, AttributeError) as e:
                self._send(400, {"error": str(e)})
                return
            try:
                responses = engine.submit(prompts, max_tokens)
            except Exception as e:              # noqa: BLE001
                self._send(500, {"error": f"{type(e).__name__}: {e}"})
                return
            self._send(200, {"model": MODEL_NAME, "responses": responses})

        def log_message(self, *args):           # quiet
            pass

    return CompletionHandler


def serve(generate_fn, port=8000, host="", max_batch=64):
    """-> (httpd, engine); call httpd.serve_forever() (or run it in a thread) and engine.close() at the end."""
    engine = EngineQueue(generate_fn, max_batch=max_batch)
    httpd = ThreadingHTTPServer((host, port), make_handler(engine))
    return httpd, engine


def run(port=8000, synthetic=False, blind_model=False):
    from .api import generate, load
    preload = load(blind_model=blind_model, synthetic=synthetic or None)

    def generate_fn(prompts, max_tokens):
        return generate(prompts if len(prompts) > 1 else prompts[0], preload=preload, max_tokens=max_tokens, verbose=False)

    httpd, engine = serve(generate_fn, port=port)
    print(f"Starting server on port {port}")
    try:
        httpd.serve_forever()
    finally:
        engine.close()


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--synthetic", action="store_true", help="seeded random weights instead of models/phi3_v")
    ap.add_argument("--blind", action="store_true", help="text-only Phi-3-mini-128K")
    a = ap.parse_args()
    run(a.port, a.synthetic, a.blind)
