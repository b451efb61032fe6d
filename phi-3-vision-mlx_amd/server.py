"""HTTP façade over `generate` -- the endpoint contract of the reference's server.py (server.py:8-29):

    POST /v1/completions   {"prompt": str | [str, ...], "max_tokens": int (default 512)}
      -> 200 {"model": "phi-3-vision", "responses": [str, ...]}            anything else -> 404
    (extension: "images": [null | "data:image/...;base64,..." per prompt] -- the reference's endpoint is text-only.
     Only INLINE images by default: a path or URL in a request body would let any client make the server open local files
     or fetch arbitrary URLs.  `ImagePolicy(allow_dir=..., allow_hosts=...)` / `--image-dir` / `--image-host` opt in to an
     allow-listed directory / host list.  Images are always fetched AND decoded in the HTTP handler thread, with a timeout,
     a byte cap and a pixel cap -- never on the engine thread -- and errors never echo the path.)

with one difference in the plumbing: requests do not call the model from the HTTP thread.  They go into a queue that a
single engine thread drains (the model object holds one in-flight sequence group).  By default every
request is its own `generate` call, as in the reference's server -- a client's output never depends on who else is in
flight.  `merge=True` (opt-in, `--merge`) folds requests that wait at the same time and ask for the same `max_tokens`
into ONE batched call: throughput for left-pad geometry that now depends on the longest co-batched prompt (results of
valid rows are pad-invariant EXCEPT the one-shot short/long RoPE choice, phi.py:492 -- so requests are only merged
while prompt + max_tokens of the merged batch stays on the same side of the 4096-token window as each request alone;
`regime_fn` supplies the prompt lengths).  `max_tokens` is clamped to `max_tokens_cap` (an unbounded value would size
the KV cache).  Malformed bodies get 400 instead of a dropped connection; a request that waits longer than `timeout_s`
gets 500.

`--continuous` replaces the queue by `engine.ContinuousEngine`: requests are prefilled into free batch rows between the
decode steps of the rows already generating and leave at their own EOS / budget (no request waits for a batch to drain).

    python -m phi_3_vision_mlx_amd.server --port 8000 [--synthetic] [--blind] [--merge | --continuous]
"""
import json
import queue
import os
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

MODEL_NAME = "phi-3-vision"


class _Job:
    __slots__ = ("prompts", "max_tokens", "images", "done", "result", "error")

    def __init__(self, prompts, max_tokens, images=None):
        self.prompts, self.max_tokens, self.images = prompts, max_tokens, images
        self.done, self.result, self.error = threading.Event(), None, None


class ImagePolicy:
    """What the `images` field of a request may name.  Default: inline `data:` URIs only."""

    def __init__(self, allow_dir=None, allow_hosts=(), max_bytes=16 << 20, max_pixels=64 << 20, timeout_s=5.0):
        import os
        self.allow_dir = os.path.realpath(allow_dir) if allow_dir else None
        self.allow_hosts = frozenset(h.lower() for h in allow_hosts)
        self.max_bytes, self.max_pixels, self.timeout_s = int(max_bytes), int(max_pixels), float(timeout_s)


def _open_image(raw, policy):
    """bytes -> a fully decoded RGB PIL image (decoded HERE, in the caller's thread), under the policy's caps."""
    from io import BytesIO
    from PIL import Image
    if len(raw) > policy.max_bytes:
        raise ValueError("image larger than the server's byte limit")
    try:
        im = Image.open(BytesIO(raw))
        if im.width * im.height > policy.max_pixels:
            raise ValueError("image larger than the server's pixel limit")
        im.load()
        return im.convert("RGB")
    except ValueError:
        raise
    except Exception:                           # noqa: BLE001 -- PIL raises many types; none of them is the client's business
        raise ValueError("image could not be decoded") from None


def decode_image(spec, policy=None):
    """One entry of a request's `images` list -> None or a decoded PIL image.  ValueError (-> HTTP 400) for anything the
    policy does not allow; messages never contain the path / URL (no file-existence oracle)."""
    policy = policy or ImagePolicy()
    if spec is None:
        return None
    if not isinstance(spec, str):
        raise ValueError("images entries must be null or strings")
    if spec.startswith("data:"):
        import base64
        import binascii
        head, _, payload = spec.partition(",")
        if not head.endswith(";base64") or len(payload) > policy.max_bytes * 4 // 3 + 4:
            raise ValueError("images: expected a base64 data URI within the size limit")
        try:
            raw = base64.b64decode(payload, validate=True)
        except (binascii.Error, ValueError):
            raise ValueError("images: invalid base64 payload") from None
        return _open_image(raw, policy)
    if spec.startswith(("http://", "https://")):
        from urllib.parse import urlsplit
        host = (urlsplit(spec).hostname or "").lower()
        if host not in policy.allow_hosts:
            raise ValueError("images: URLs are not accepted by this server (inline data: URIs only)")
        import requests
        try:
            with requests.get(spec, stream=True, timeout=policy.timeout_s, allow_redirects=False) as resp:
                resp.raise_for_status()
                raw = resp.raw.read(policy.max_bytes + 1, decode_content=True)
        except Exception:                       # noqa: BLE001
            raise ValueError("images: fetch failed") from None
        return _open_image(raw, policy)
    if policy.allow_dir is None:
        raise ValueError("images: file paths are not accepted by this server (inline data: URIs only)")
    import os
    path = os.path.realpath(os.path.join(policy.allow_dir, spec))
    if os.path.commonpath([path, policy.allow_dir]) != policy.allow_dir or not os.path.isfile(path):
        raise ValueError("images: not an image of the served directory")
    if os.path.getsize(path) > policy.max_bytes:
        raise ValueError("image larger than the server's byte limit")
    with open(path, "rb") as f:
        return _open_image(f.read(), policy)


class EngineQueue:
    """Single consumer in front of a non-re-entrant `generate_fn(prompts: list[str], max_tokens) -> str | list[str]`."""

    def __init__(self, generate_fn, max_batch=64, window_s=0.005, merge=False, max_tokens_cap=4096, timeout_s=600.0,
                 length_fn=None, window_tokens=4096, device=None):
        self.generate_fn, self.max_batch, self.window_s = generate_fn, max_batch, window_s
        self.merge, self.max_tokens_cap, self.timeout_s = merge, max_tokens_cap, timeout_s
        self.length_fn, self.window_tokens, self.device = length_fn, window_tokens, device
        self.jobs = queue.Queue()
        self.batches = []                       # sizes of the generate calls issued (observability / tests)
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, prompts, max_tokens, images=None):
        job = _Job(prompts, max(1, min(int(max_tokens), self.max_tokens_cap)), images)
        self.jobs.put(job)
        if not job.done.wait(self.timeout_s):
            raise TimeoutError(f"no result within {self.timeout_s} s")
        if job.error is not None:
            raise job.error
        return job.result

    def close(self):
        self._stop = True
        self.jobs.put(None)
        self.thread.join(timeout=5)

    def _regime(self, prompts, max_tokens):
        """Side of the RoPE window the batch falls on (phi.py:492: long factors iff longest prompt + max_tokens > 4096)."""
        if self.length_fn is None:
            return None
        return max(self.length_fn(p) for p in prompts) + max_tokens > self.window_tokens

    def _collect(self, first):
        """merge=False: `first` alone.  merge=True: plus every queued job with the same max_tokens that fits and keeps the
        RoPE regime of each member unchanged, waiting at most `window_s` for stragglers."""
        group, n, held = [first], len(first.prompts), []
        if not self.merge or first.images is not None:          # image requests are never merged
            return group
        regime = self._regime(first.prompts, first.max_tokens)
        while n < self.max_batch:
            try:
                job = self.jobs.get(timeout=self.window_s)
            except queue.Empty:
                break
            if job is None:
                self.jobs.put(None)
                break
            if job.images is None and job.max_tokens == first.max_tokens and n + len(job.prompts) <= self.max_batch \
                    and self._regime(job.prompts, job.max_tokens) == regime:
                group.append(job)
                n += len(job.prompts)
            else:
                held.append(job)
        for job in held:                        # different budget: next round, order kept
            self.jobs.put(job)
        return group

    def _run(self):
        if self.device is not None:                             # a new thread starts on GPU 0 whatever the loader used
            import torch
            torch.cuda.set_device(self.device)
        while not self._stop:
            first = self.jobs.get()
            if first is None:
                break
            group = self._collect(first)
            flat = [p for j in group for p in j.prompts]
            try:
                out = self.generate_fn(flat, first.max_tokens) if first.images is None else \
                    self.generate_fn(flat, first.max_tokens, first.images)
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


def make_handler(engine, image_policy=None):
    image_policy = image_policy or ImagePolicy()

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
                images = request.get("images")
                if images is not None:
                    images = [images] if isinstance(images, str) else list(images)
                    if len(images) != len(prompts):
                        raise ValueError("images must list one entry (or null) per prompt")
                    images = [decode_image(i, image_policy) for i in images]      # fetched + decoded in THIS thread
                    images = None if all(i is None for i in images) else images
            except (ValueError, TypeError, AttributeError, OSError) as e:
                self._send(400, {"error": str(e)})
                return
            try:
                responses = engine.submit(prompts, max_tokens, images) if images is not None else engine.submit(prompts, max_tokens)
            except Exception as e:              # noqa: BLE001
                self._send(500, {"error": f"{type(e).__name__}: {e}"})
                return
            self._send(200, {"model": MODEL_NAME, "responses": responses})

        def log_message(self, *args):           # quiet
            pass

    return CompletionHandler


def serve(generate_fn, port=8000, host="127.0.0.1", max_batch=64, image_policy=None, **engine_kwargs):
    """-> (httpd, engine); call httpd.serve_forever() (or run it in a thread) and engine.close() at the end.
    Binds the loopback interface unless told otherwise (`host=""` = all interfaces, as the reference's server.py:31)."""
    engine = EngineQueue(generate_fn, max_batch=max_batch, **engine_kwargs)
    httpd = ThreadingHTTPServer((host, port), make_handler(engine, image_policy))
    return httpd, engine


class ContinuousBackend:
    """The handler's `submit` on top of engine.ContinuousEngine (its own stepping thread)."""

    def __init__(self, engine, max_tokens_cap=4096, timeout_s=600.0):
        self.engine, self.max_tokens_cap, self.timeout_s = engine, max_tokens_cap, timeout_s
        self.stop = threading.Event()
        self.thread = threading.Thread(target=engine.serve_forever, args=(self.stop,), daemon=True)
        self.thread.start()

    def submit(self, prompts, max_tokens, images=None):
        return self.engine.generate(prompts, images, max(1, min(int(max_tokens), self.max_tokens_cap)), self.timeout_s)

    def close(self):
        self.stop.set()
        self.thread.join(timeout=5)


def serve_continuous(engine, port=8000, host="127.0.0.1", image_policy=None, **kw):
    backend = ContinuousBackend(engine, **kw)
    return ThreadingHTTPServer((host, port), make_handler(backend, image_policy)), backend


def run(port=8000, synthetic=False, blind_model=False, merge=False, continuous=False, host="127.0.0.1", image_policy=None,
        long_window=0, slots=8):
    from .api import _apply_chat_template, generate, load
    preload = load(blind_model=blind_model, synthetic=synthetic or None)
    processor = preload[1]
    preload[0].serving = True            # a server does not own the GPU: no launch that needs its whole grid resident at once (model.py)
    if continuous:
        import torch.distributed as dist
        from .engine import ContinuousEngine, RegimeRouter
        eng = ContinuousEngine(*preload, slots=slots)                 # requests with prompt + max_tokens <= 4096 (short RoPE factors)
        if long_window > 4096:                                       # + one engine for the long-RoPE regime (phi.py:492)
            eng = RegimeRouter([eng, ContinuousEngine(*preload, slots=max(1, slots // 2), window=long_window)])
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        front = None
        if world > 1:                                                # one engine per rank (= per GPU), rank 0 dispatches (fleet.py)
            from . import fleet
            groups = fleet.make_groups()
            if dist.get_rank() != 0:
                fleet.worker(eng, groups)
                return
            eng = front = fleet.EngineFleet(eng, groups, world)
        httpd, engine = serve_continuous(eng, port=port, host=host, image_policy=image_policy)
        print(f"Starting server on port {port} (continuous batching{f', {world} engines' if world > 1 else ''})")
        try:
            httpd.serve_forever()
        finally:
            engine.close()
            if front is not None:
                front.close()
        return

    def generate_fn(prompts, max_tokens, images=None):
        import torch.distributed as dist
        if images is not None or (dist.is_available() and dist.is_initialized() and len(prompts) > 1):
            # mixed image + text requests -- and text-only batches whenever a process group exists -- go through the
            # batch-sharded path (dist.py: one left-padded batch per rank; world size 1 = this GPU alone)
            from .dist import generate_sharded
            return generate_sharded(prompts, images if images is not None else [None] * len(prompts), preload=preload,
                                    max_tokens=max_tokens)
        return generate(prompts if len(prompts) > 1 else prompts[0], preload=preload, max_tokens=max_tokens, verbose=False)

    def length_fn(prompt):
        return len(processor.tokenizer(_apply_chat_template(prompt, None, False)[0]).input_ids)

    httpd, engine = serve(generate_fn, port=port, host=host, merge=merge, length_fn=length_fn, device=preload[0].device,
                          image_policy=image_policy)
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
    ap.add_argument("--tiny", action="store_true", help="with --synthetic: the 2-layer test model (smoke tests)")
    ap.add_argument("--blind", action="store_true", help="text-only Phi-3-mini-128K")
    ap.add_argument("--merge", action="store_true", help="fold concurrent same-budget requests into one batched generate (opt-in)")
    ap.add_argument("--continuous", action="store_true", help="continuous batching engine (requests join / leave between decode steps)")
    ap.add_argument("--long-window", type=int, default=0, help="with --continuous: also serve prompt + max_tokens > 4096 up to this many tokens")
    ap.add_argument("--slots", type=int, default=8, help="with --continuous: batch rows of the engine")
    ap.add_argument("--host", default="127.0.0.1", help='interface to bind ("" = all, as the reference)')
    ap.add_argument("--image-dir", default=None, help="allow `images` entries naming files under this directory")
    ap.add_argument("--image-host", action="append", default=[], help="allow `images` URLs on this host (repeatable)")
    a = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:               # python -m torch.distributed.run --nproc-per-node N -m ...server --continuous
        if not a.continuous:
            # the one-shot paths have no worker loop: every rank would bind the same port and rank 0's batches would wait in
            # generate_sharded for ranks that never join
            ap.error("WORLD_SIZE > 1 needs --continuous (one engine per rank, rank 0 serves HTTP); "
                     "for one-shot batch sharding call dist.generate_sharded from a torchrun script instead")
        import torch
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())   # (fewer GPUs than ranks: shared)
        dist.init_process_group("gloo")                          # requests and token lists only: host memory (fleet.py)
    run(a.port, "tiny" if a.synthetic and a.tiny else a.synthetic, a.blind, a.merge, a.continuous, a.host, ImagePolicy(a.image_dir, a.image_host), a.long_window, a.slots)
