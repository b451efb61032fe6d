"""The HTTP façade: reference endpoint contract (server.py:8-29) + request queue + batching."""
import json
import threading
import time
import urllib.error
import urllib.request

import pytest


def _post(port, path, payload, raw=None):
    data = raw if raw is not None else json.dumps(payload).encode()
    req = urllib.request.Request(f"http://127.0.0.1:{port}{path}", data=data, headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=10) as r:
        return r.status, json.loads(r.read())


@pytest.fixture(params=[True, False], ids=["merge", "solo"])
def server(request):
    from phi_3_vision_mlx_amd.server import serve
    calls = []

    def fake_generate(prompts, max_tokens):
        calls.append((list(prompts), max_tokens))
        time.sleep(0.05)                                     # long enough for concurrent requests to pile up
        if any(p == "boom" for p in prompts):
            raise RuntimeError("engine failure")
        out = [f"{p}|{max_tokens}" for p in prompts]
        return out[0] if len(out) == 1 else out              # like generate(): str for B=1, list otherwise

    # length_fn: prompts starting with "L" count as 4000 tokens (the far side of the RoPE window with any budget >= 97)
    httpd, engine = serve(fake_generate, port=0, host="127.0.0.1", merge=request.param, max_tokens_cap=1000,
                          length_fn=lambda p: 4000 if p.startswith("L") else len(p))
    engine.merge_param = request.param
    t = threading.Thread(target=httpd.serve_forever, daemon=True)
    t.start()
    yield httpd.server_address[1], calls, engine
    httpd.shutdown()
    engine.close()


def test_completions_contract(server):
    port, calls, _ = server
    assert _post(port, "/v1/completions", {"prompt": "Hello, world!", "max_tokens": 50}) == \
        (200, {"model": "phi-3-vision", "responses": ["Hello, world!|50"]})
    assert _post(port, "/v1/completions", {"prompt": ["a", "b"]})[1]["responses"] == ["a|512", "b|512"]   # default budget
    with pytest.raises(urllib.error.HTTPError) as e:
        _post(port, "/v1/other", {"prompt": "x"})
    assert e.value.code == 404
    for bad in (b"{not json", json.dumps({"prompt": 7}).encode(), json.dumps({"prompt": []}).encode()):
        with pytest.raises(urllib.error.HTTPError) as e:
            _post(port, "/v1/completions", None, raw=bad)
        assert e.value.code == 400
    with pytest.raises(urllib.error.HTTPError) as e:
        _post(port, "/v1/completions", {"prompt": "boom"})
    assert e.value.code == 500
    assert _post(port, "/v1/completions", {"prompt": "still alive", "max_tokens": 3})[0] == 200


def test_concurrent_requests_are_batched_and_routed_back(server):
    port, calls, engine = server
    results = {}

    def worker(i, mt):
        results[i] = _post(port, "/v1/completions", {"prompt": [f"p{i}a", f"p{i}b"], "max_tokens": mt})[1]["responses"]

    _post(port, "/v1/completions", {"prompt": "warm"})        # engine busy -> the next ones queue up behind it
    ths = [threading.Thread(target=worker, args=(i, 8 if i < 4 else 9)) for i in range(6)]
    blocker = threading.Thread(target=worker, args=(99, 7))
    blocker.start()
    time.sleep(0.01)
    [t.start() for t in ths]
    [t.join() for t in ths + [blocker]]
    for i in range(6):
        mt = 8 if i < 4 else 9
        assert results[i] == [f"p{i}a|{mt}", f"p{i}b|{mt}"]      # every request got exactly its own texts
    if engine.merge_param:
        assert max(engine.batches) >= 4                         # opt-in: same-budget requests were merged into one call
    else:
        assert max(engine.batches) == 2                         # default: every request is its own generate call
    want_mt = {f"p{i}{s}": (8 if i < 4 else 9) for i in range(6) for s in "ab"} | {"p99a": 7, "p99b": 7, "warm": 512}
    for prompts, mt in calls:                                   # a batch never mixes budgets
        assert all(want_mt[p] == mt for p in prompts), (prompts, mt)


def test_max_tokens_is_clamped_and_regimes_do_not_mix(server):
    port, calls, engine = server
    assert _post(port, "/v1/completions", {"prompt": "x", "max_tokens": 10 ** 9})[1]["responses"] == ["x|1000"]
    results = {}

    def worker(p):
        results[p] = _post(port, "/v1/completions", {"prompt": p, "max_tokens": 200})[1]["responses"]
    _post(port, "/v1/completions", {"prompt": "warm"})
    ths = [threading.Thread(target=worker, args=(p,)) for p in ("Long one", "short a", "short b", "Long two")]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(results[p] == [f"{p}|200"] for p in results) and len(results) == 4
    for prompts, mt in calls:                                   # a long-RoPE request never shares a batch with a short one
        assert len({p.startswith("L") for p in prompts}) == 1, prompts


def _png_data_uri(w=8, h=8):
    import base64
    from io import BytesIO
    import numpy as np
    from PIL import Image
    buf = BytesIO()
    Image.fromarray(np.full((h, w, 3), 200, dtype=np.uint8)).save(buf, format="PNG")
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()


def test_image_policy_inline_only_by_default(tmp_path):
    """the HTTP `images` field must not make the server open local paths or fetch URLs (SSRF / file oracle)."""
    from PIL import Image
    from phi_3_vision_mlx_amd.server import ImagePolicy, decode_image
    secret = tmp_path / "secret.png"
    Image.new("RGB", (4, 4)).save(secret)
    assert decode_image(None) is None
    im = decode_image(_png_data_uri())
    assert im.size == (8, 8) and im.mode == "RGB"
    for spec in (str(secret), "/etc/passwd", "/nonexistent/x.png", "http://169.254.169.254/latest/meta-data", "https://example.com/a.png"):
        with pytest.raises(ValueError) as e:
            decode_image(spec)
        assert spec not in str(e.value) and "secret" not in str(e.value)      # nothing about the target leaks
    # an existing and a missing file are indistinguishable from outside
    with pytest.raises(ValueError) as e1:
        decode_image(str(secret))
    with pytest.raises(ValueError) as e2:
        decode_image(str(tmp_path / "missing.png"))
    assert str(e1.value) == str(e2.value)
    for bad in (7, ["x"], {"a": 1}):
        with pytest.raises(ValueError):
            decode_image(bad)
    with pytest.raises(ValueError):
        decode_image("data:image/png;base64,!!!notbase64!!!")
    with pytest.raises(ValueError):
        decode_image("data:image/png;base64," + "QUJD" * 10)                 # valid base64, not an image
    with pytest.raises(ValueError):
        decode_image(_png_data_uri(64, 64), ImagePolicy(max_pixels=1000))
    with pytest.raises(ValueError):
        decode_image(_png_data_uri(), ImagePolicy(max_bytes=16))
    # opt-in directory: files inside it load, traversal out of it does not
    pol = ImagePolicy(allow_dir=str(tmp_path))
    assert decode_image("secret.png", pol).size == (4, 4)
    outside = tmp_path.parent / "outside.png"
    Image.new("RGB", (4, 4)).save(outside)
    for spec in ("../outside.png", str(outside), "/etc/passwd"):
        with pytest.raises(ValueError):
            decode_image(spec, pol)
    # opt-in hosts: any other host is still refused before a connection is made
    with pytest.raises(ValueError, match="not accepted"):
        decode_image("http://evil.example/x.png", ImagePolicy(allow_hosts=["images.example"]))


def test_http_images_field_is_validated_in_the_handler():
    from phi_3_vision_mlx_amd.server import serve
    seen = []

    def fake_generate(prompts, max_tokens, images=None):
        seen.append(images)
        return [f"{p}|{'img' if im is not None else 'txt'}" for p, im in zip(prompts, images or [None] * len(prompts))]

    httpd, engine = serve(fake_generate, port=0)
    assert httpd.server_address[0] == "127.0.0.1"                           # loopback unless asked otherwise
    threading.Thread(target=httpd.serve_forever, daemon=True).start()
    port = httpd.server_address[1]
    try:
        ok = _post(port, "/v1/completions", {"prompt": ["a", "b"], "images": [_png_data_uri(), None]})
        assert ok == (200, {"model": "phi-3-vision", "responses": ["a|img", "b|txt"]})
        assert seen[-1][0].size == (8, 8) and seen[-1][1] is None           # the engine got a DECODED image, not a string
        for images in (["/etc/passwd"], ["http://127.0.0.1:1/x.png"], [7], [["x"]]):
            with pytest.raises(urllib.error.HTTPError) as e:
                _post(port, "/v1/completions", {"prompt": ["a"], "images": images})
            assert e.value.code == 400
            assert "passwd" not in e.value.read().decode()
        assert len(seen) == 1                                               # none of the bad requests reached the engine
    finally:
        httpd.shutdown()
        engine.close()
