"""Public inference API -- the drop-in boundary.

Same names, signatures, defaults, return types and error behaviour as the
reference's `load / generate / choose / constrain / benchmark`
(reference phi_3_vision_mlx.py:1178-1487) and the decoding loops underneath
(`_generate :376-409`, `_choose_from :466-487`, `_constrain :500-619`,
`Streamer / LogitStopper / TokenStopper :45-117`, `_apply_chat_template :341-357`).
Host code stays Python; everything that touches the device goes through
`model(...)` (model.py) and `model_ops` (ops.py), i.e. hand-written HIP kernels.

`quantize_cache=True` selects the int8 KV cache and `quantize_model=True` fp8 (e4m3) decoder weights (or the reference's
own 4-bit group-64 format: an MLX `*_Q` checkpoint directory if present, or `quantize_format="int4"`) -- the build's
analogues of the reference's 4-bit prompt cache / int4 weights (BASELINE config 5).
`use_adapter=True` attaches the LoRA adapter of `adapters/<model dir name>` (or `adapter_path=...`) at inference
(reference phi_3_vision_mlx.py:266-271); training adapters is not part of this build.
Not carried over (out of scope): the `<|api_input|>` tool hook,
HF-hub download (`_setup`): a model directory must exist locally, or pass
`synthetic=...` to `load()` for seeded random weights of the real architecture.
"""
import json
import math
import os
import sys
import time
from io import BytesIO
from pathlib import Path

import numpy as np
import torch

from . import ops as model_ops
from .config import is_vision, load_config, make_config, phi3v_config_dict, tiny_config_dict
from .processor import Phi3FProcessor, Phi3VProcessor
from .weights import load_adapter, load_safetensors_dir, resolve_adapter, synth_weights

PATH_ADAPTERS = "adapters"
PATH_ORIGINAL_PHI3_VISION = "models/phi3_v"
PATH_QUANTIZED_PHI3_VISION = "models/phi3_v_Q"
PATH_ORIGINAL_PHI3_BLIND = "models/phi3_mini_128k"
PATH_QUANTIZED_PHI3_BLIND = "models/phi3_mini_128k_Q"
ID_EOS = 32007
ID_ASS = 32001


class Tic:
    """reference phi.py:16-24."""

    def __init__(self):
        self.last_time = time.perf_counter()

    def __call__(self):
        now = time.perf_counter()
        dt, self.last_time = now - self.last_time, now
        return dt


def _rows(token):
    """token: int tensor [B,1] / [B] or nested list -> list[int] per row (one D2H copy).  A negative id is the device's
    report of a failed step (NaN logits, see p3v_argmax / p3v_step_end): raise instead of decoding garbage."""
    rows = token.reshape(-1).tolist() if torch.is_tensor(token) else np.asarray(token).reshape(-1).tolist()
    if rows and min(rows) < 0:
        raise RuntimeError(f"device step failed: NaN logits (token ids {rows})")
    return rows


class _StepFailed(RuntimeError):
    """a decode step delivered a negative token id (NaN logits, or a poisoned row)"""


class Streamer:
    """reference phi_3_vision_mlx.py:45-77."""

    def __init__(self, processor, stream, mute):
        self.tokenizer = processor.tokenizer
        self.mute = mute
        self.stream = stream and (not mute)
        self.list_tokens = []
        self.idx_sofar = 0

    def __call__(self, token):
        rows = _rows(token)
        if not self.stream:
            self.list_tokens.append(rows)
            return None
        if len(rows) > 1:
            self.list_tokens.append(rows)
            self.stream = False
            return None
        self.list_tokens.append(rows[0])
        txt = self.tokenizer.decode(self.list_tokens)
        idx_split = txt.rfind(" ", self.idx_sofar)
        if idx_split > 0:
            print(txt[self.idx_sofar:idx_split], end="", flush=True)
            self.idx_sofar = idx_split

    def end(self):
        if self.stream:
            txt = self.tokenizer.decode(self.list_tokens)
            print(txt[self.idx_sofar:], "\n", flush=True)
            return txt, len(self.list_tokens)
        steps = [s if isinstance(s, list) else [s] for s in self.list_tokens]
        per_row = [list(r) for r in zip(*steps)]                  # [B][n_steps]
        list_txt = self.tokenizer.batch_decode([(r[:r.index(ID_EOS) + 1] if ID_EOS in r else r) for r in per_row])
        if not self.mute:
            for i, gen in enumerate(list_txt):
                print(f"\n< Generated text for prompt #{i} >\n{gen}")
        return list_txt, sum(len(r) for r in per_row)


class LogitStopper:
    """Early-stop heuristic, B=1 only (reference phi_3_vision_mlx.py:79-104)."""

    def __init__(self, max_tokens, early_stop):
        self.step = 0
        # isinstance(True, int) holds in Python: the reference takes early_stop=True as the integer 1 (:82) -- kept
        self.early_stop = early_stop if isinstance(early_stop, int) and (early_stop < max_tokens) else False
        self.log_prob_sum = 0.0
        self.best_eos_sofar = -math.inf
        self.log_prob_sum_at_best_eos = 0.0

    def __call__(self, logits):
        if not self.early_stop:
            return False
        if logits.shape[0] > 1:
            self.early_stop = False
            return False
        log_prob = model_ops.log_softmax(logits[:, -1, :].contiguous())
        log_prob_best = log_prob.float().max().item()
        log_prob_eos = log_prob[0, ID_EOS].float().item()
        if log_prob_eos > self.best_eos_sofar:
            self.log_prob_sum_since_last_best_eos = self.log_prob_sum - self.log_prob_sum_at_best_eos
            if (self.log_prob_sum_since_last_best_eos < self.best_eos_sofar) and (self.step > self.early_stop):
                return True
            self.best_eos_sofar = log_prob_eos
            self.log_prob_sum_at_best_eos = self.log_prob_sum
        self.log_prob_sum += log_prob_best
        self.step += 1
        return False


class TokenStopper:
    """Stop when every row has emitted EOS (reference phi_3_vision_mlx.py:105-117)."""

    def __init__(self, processor, batch_size):
        self.tokenizer = processor.tokenizer
        self.eos_id = ID_EOS
        self.batch_size = batch_size
        self.eos_rows = [1] * batch_size

    def __call__(self, token):
        rows = _rows(token)
        if self.eos_id in rows:
            self.eos_rows = [a * int(t != self.eos_id) for a, t in zip(self.eos_rows, rows)]
            if sum(self.eos_rows) < 1:
                return True
        return False


# ----------------------------------------------------------------------------- loading
def _load_image(image_source):
    """reference phi_3_vision_mlx.py:307-326 (PIL images are passed through)."""
    from PIL import Image
    if isinstance(image_source, Image.Image):
        return image_source
    if isinstance(image_source, BytesIO):
        try:
            return Image.open(image_source)
        except IOError as e:
            raise ValueError(f"Failed to load image from BytesIO with error: {e}")
    elif image_source.startswith(("http://", "https://")):
        try:
            import requests
            response = requests.get(image_source, stream=True)
            response.raise_for_status()
            return Image.open(response.raw)
        except Exception as e:
            raise ValueError(f"Failed to load image from URL: {image_source} with error {e}")
    elif Path(image_source).is_file():
        try:
            return Image.open(image_source)
        except IOError as e:
            raise ValueError(f"Failed to load image {image_source} with error: {e}")
    else:
        raise ValueError(f"The image {image_source} must be a valid URL or existing file.")


def _apply_chat_template(prompt, images, verbose, apply_chat_template=True):
    """reference phi_3_vision_mlx.py:341-357."""
    if apply_chat_template is False:
        print(f"*** Prompt ***\n{prompt}\n*** Images ***\n{images}\n*** Output ***") if verbose else None
        return prompt, images
    if images is not None:
        images = [_load_image(i) for i in images] if isinstance(images, list) else [_load_image(images)]
        img_prompt = "\n".join([f"<|image_{i+1}|>" for i in range(len(images))]) + "\n"
    else:
        img_prompt = ""
    prompt = [prompt] if isinstance(prompt, str) else prompt
    prompt = [f"<|user|>\n{img_prompt}{i.strip()}<|end|>\n<|assistant|>\n" for i in prompt]
    if verbose:
        prompt_str = "\n".join(map(str.strip, prompt)).strip()
        images_str = "\n".join(map(str, images)) if images else "None"
        print(f"*** Prompt ***\n{prompt_str}\n*** Images ***\n{images_str}\n*** Output ***")
    prompt = prompt[0] if len(prompt) == 1 else prompt
    return prompt, images


def _get_cfg(json_path, **kwargs):
    return load_config(json_path, **kwargs)


def _make_processor(cfg, model_path, return_mx=True):
    cls = Phi3VProcessor if is_vision(cfg) else Phi3FProcessor
    return cls(model_path, return_mx=return_mx)


def _load(model_path=PATH_ORIGINAL_PHI3_VISION, adapter_path=None, return_mx=True, device=None, **kwargs):
    """reference phi_3_vision_mlx.py:257-274: config -> model class by `architectures[0]`,
    HF safetensors -> device weights."""
    from .model import Phi3VModel
    cfg = _get_cfg(f"{model_path}/config.json", **kwargs)
    processor = _make_processor(cfg, model_path, return_mx)
    device = device or f"cuda:{torch.cuda.current_device()}"
    model = Phi3VModel(cfg, load_safetensors_dir(model_path, cfg, device="cpu"), device=device)
    if adapter_path:
        _attach_adapter(model, adapter_path, model_path)
    return model, processor


def _get_adapter_path(model_path):
    """reference phi_3_vision_mlx.py:462-464."""
    print(f"{PATH_ADAPTERS}/{Path(model_path).name}")
    return f"{PATH_ADAPTERS}/{Path(model_path).name}"


def _attach_adapter(model, adapter_path, model_path=None):
    """reference phi_3_vision_mlx.py:266-271: adapter_config.json -> LoRA layers, adapters.safetensors -> their weights."""
    lora_cfg, tensors = load_adapter(adapter_path)
    if model_path is not None and lora_cfg.get("model_path") != model_path:
        print(f"WARNING: LoRA trained for {lora_cfg.get('model_path')} is being used with {model_path}")
    model.set_adapters(resolve_adapter(model.cfg, lora_cfg, tensors))


def load_synthetic(blind_model=False, tiny=False, seed=0, device=None, std_scale=1.0, adapter_path=None, lm_head_spread=0.0,
                   lm_head_seed=0, outliers=None, residual_scale=None, **kwargs):
    """Seeded random weights of the real (or tiny) architecture -- no checkpoint needed.
    lm_head_spread / lm_head_seed: decisive-argmax head of the parity fixtures (weights.peaked_lm_head);
    outliers: heavy-tailed activations (weights.add_outliers); residual_scale: the well-conditioned checkpoint (weights.synth_weights)."""
    from .model import Phi3VModel
    d = tiny_config_dict(vision=not blind_model) if tiny else phi3v_config_dict(vision=not blind_model)
    cfg = make_config(d, **kwargs)
    device = device or f"cuda:{torch.cuda.current_device()}"
    model = Phi3VModel(cfg, synth_weights(cfg, seed=seed, device=device, std_scale=std_scale, lm_head_spread=lm_head_spread,
                                          lm_head_seed=lm_head_seed, outliers=outliers, residual_scale=residual_scale), device=device)
    if adapter_path:
        _attach_adapter(model, adapter_path)
    return model, _make_processor(cfg, None)


def load(blind_model=False, quantize_model=False, quantize_cache=False, use_adapter=False, **kwargs):
    """reference phi_3_vision_mlx.py:1279-1322.  Extra: `synthetic=True|'tiny'` builds seeded
    random weights instead of reading `models/...` (there is no hub access here)."""
    synthetic = kwargs.pop("synthetic", None)
    adapter_path = kwargs.pop("adapter_path", None)
    fmt = kwargs.pop("quantize_format", "fp8")               # on-the-fly weight format of quantize_model=True: "fp8" | "int4"
    if fmt not in ("fp8", "int4"):
        raise ValueError(f"quantize_format must be 'fp8' or 'int4', got {fmt!r}")
    model_path = kwargs.pop("model_path", None)
    if model_path is None:
        model_path = PATH_ORIGINAL_PHI3_BLIND if blind_model else PATH_ORIGINAL_PHI3_VISION
        q_path = PATH_QUANTIZED_PHI3_BLIND if blind_model else PATH_QUANTIZED_PHI3_VISION
        if quantize_model and not synthetic and os.path.exists(q_path):
            model_path, quantize_model = q_path, False          # reference :1305-1311: an MLX 4-bit checkpoint, loaded as is
    if use_adapter and adapter_path is None:
        adapter_path = _get_adapter_path(model_path)                                   # reference :1316-1317
    if synthetic:
        return load_synthetic(blind_model=blind_model, tiny=(synthetic == "tiny"), use_quantized_cache=quantize_cache,
                              quantized_fp8=quantize_model and fmt == "fp8", quantized_int4=quantize_model and fmt == "int4",
                              adapter_path=adapter_path if use_adapter else None, **kwargs)
    if not os.path.exists(model_path):
        raise FileNotFoundError(
            f"model directory {model_path!r} not found and this build cannot download checkpoints; "
            "place HF-layout safetensors + config.json there, or use load(synthetic=True)")
    return _load(model_path=model_path, use_quantized_cache=quantize_cache, quantized_fp8=quantize_model and fmt == "fp8",
                 quantized_int4=quantize_model and fmt == "int4",
                 adapter_path=adapter_path if use_adapter else None, **kwargs)


# ----------------------------------------------------------------------------- generate
def _last_logits(logits):
    return logits[:, -1, :].contiguous()


def greedy_loop(model, token, cache, n_steps, streamer, token_stopper, logit_stopper=None, mask=None, pids=None):
    """The decode loop of `_generate` (reference phi_3_vision_mlx.py:390-398): `n_steps` greedy steps after the prefill token,
    every token handed to the streamer and the stoppers in order, stops as the reference stops.

    With the graph-replayed step and no logit stopper the loop is ONE STEP AHEAD of the host: step i + 1 is enqueued (its input
    token never leaves the device) before the host reads token i, so the per-token work the reference does after `mx.eval` --
    D2H copy, Streamer, TokenStopper -- runs while the GPU computes the next step.  Tokens, texts and stop step are identical;
    when a stop fires one speculative step has been enqueued: it is dropped -- the cache offset is rewound past it and the
    STOP step's token is returned, so cache and return value are what the one-sync-per-token loop leaves.  With a logit stopper (it reads step i's logits
    on the host before deciding) or an eager model the loop is the reference's, one sync per token."""
    graph_step = getattr(model, "greedy_step", None)
    ahead = graph_step is not None and (logit_stopper is None or not logit_stopper.early_stop) and torch.is_tensor(token) and token.is_cuda
    if not ahead:
        for _ in range(n_steps):
            if graph_step is not None:
                logits, token = graph_step(token, cache)
            else:
                logits, cache = model(input_ids=token, cache=cache, mask=mask, pids=pids)
                token = model_ops.argmax(_last_logits(logits))[:, None]
            rows = _rows(token)                                     # ONE D2H copy per step, shared by the streamer and the stopper
            streamer(rows)
            if logit_stopper is not None and logit_stopper(logits):
                break
            if token_stopper(rows):
                break
        return token
    B = token.shape[0]
    host = []                                                       # pinned staging buffers of the copy fallback (rarely needed)
    pending = None                                                  # (event, pinned buffer) of the step the host has not read yet

    debug = os.environ.get("P3V_DEBUG_STEP") == "1"
    fail_at = int(os.environ.get("P3V_DEBUG_FAIL_STEP", "-1"))      # tests: report this step's tokens as failed, once
    taken = []
    n_done = 0                                                      # steps of this call whose tokens reached the streamer

    def take(p):
        nonlocal n_done, fail_at
        p[0].synchronize()
        if debug and p[2] is not None:                              # the pinned-history read trusts n_replays == the device's step
            g_, k_ = p[2]                                           # counter; a direct graph.launch() or a d_step reset breaks that
            assert int(g_["d_step"].item()) >= k_ + 1, f"history column {k_} read, device step counter {int(g_['d_step'].item())}"
        rows = p[1].tolist()
        if n_done == fail_at:                                       # (tests) what a poisoned step leaves: a negative token, NaN cache rows
            rows, fail_at = [-1] * len(rows), -1
            torch.cuda.synchronize()
            if not st.quantized:
                st.v[..., st.offset - (i - n_done):st.offset] = float("nan")
                st.k[:, :, :, st.offset - (i - n_done):st.offset] = float("nan")
        if min(rows) < 0:
            raise _StepFailed(f"device step failed: NaN logits (token ids {rows})")
        taken[:] = rows
        n_done += 1
        streamer(rows)
        return token_stopper(rows)
    st = cache[0].state
    token0 = token.clone()                                          # (the caller's tensor may be the step's own output buffer)
    degraded = False
    i = 0
    while True:
        try:
            while i < n_steps:
                _, token = graph_step(token, cache)                 # enqueue step i
                i += 1
                g = st.graphs["greedy"]
                k, hist = g["n_replays"] - 1, g["history"]
                ev = torch.cuda.Event()
                if k < hist.shape[1] and not hist.is_cuda:
                    # the step wrote its tokens into pinned host memory itself (history[:, k], model._build_decode_graph): nothing to
                    # copy, the replays run back to back (an in-line D2H copy node costs the step ~18 us of idle GPU: 550 -> 556 tok/s)
                    ev.record()
                    src, chk = hist[:, k], (g, k)
                else:                                               # (history full, or kept on the device by another model class)
                    if not host:
                        host = [torch.empty((B,), dtype=torch.int32).pin_memory() for _ in range(2)]
                    host[i & 1].copy_(token.reshape(-1), non_blocking=True)   # stream-ordered copy, before the next replay overwrites it
                    ev.record()
                    src, chk = host[i & 1], None
                prev, pending = pending, (ev, src, chk)
                if prev is not None and take(prev):                 # host work of step i - 2 under the GPU's step i - 1
                    st.offset -= 1                                  # drop the speculative step: its K/V row lies beyond the offset
                    return torch.tensor(taken, dtype=torch.int32, device=token.device).view(-1, 1)
            if pending is not None:
                last, pending = pending, None
                take(last)
            return token
        except _StepFailed:
            # A step delivered no token.  The one launch of the step that depends on the rest of the GPU is the fused attention +
            # o_proj (every workgroup resident at once: another process on the same GPU can starve it until its bound runs out and
            # the row is poisoned).  Once per call: plan without it (as a server-owned model does), rewind to the last good token,
            # go on.  Anything else -- or a second failure -- is raised.
            g = st.graphs.get("greedy")
            if degraded or g is None or not g["bufs"].get("fuse_o", False):
                st.mark_dirty() if hasattr(st, "mark_dirty") else None    # (a captured-prefill entry on these buffers is dropped)
                raise
            degraded = True
            torch.cuda.synchronize()
            model.serving = True
            st.graphs.clear()
            st.offset -= i - n_done                                 # the failed step and the speculative one behind it
            st.scrub(st.offset, st.offset + i - n_done)             # (their cache rows hold NaN: 0 x NaN would poison every later step)
            i, pending = n_done, None
            token = torch.tensor(taken, dtype=torch.int32, device=token0.device).view(-1, 1) if n_done else token0
            print("[phi3v] a decode step timed out in the fused attention + o_proj launch (is another process using this GPU?): "
                  "continuing with separate launches", file=sys.stderr)


def _generate(model, processor, prompt, images=None, max_tokens=512, verbose=True, return_tps=False, early_stop=False,
              stream=True, mute=False):
    """Greedy decoding loop (reference phi_3_vision_mlx.py:376-409)."""
    if images is not None and isinstance(prompt, list):
        raise ValueError("Images cannot be provided when prompt is a list")
    logit_stopper = LogitStopper(max_tokens, early_stop)
    streamer = Streamer(processor, stream, mute)
    dict_input = processor(prompt, images)
    mask, pids = dict_input.get("mask", None), dict_input.get("pids", None)
    token_stopper = TokenStopper(processor, dict_input["input_ids"].shape[0])
    tic = Tic()
    logits, cache = model(**dict_input, max_tokens=max_tokens)
    token = model_ops.argmax(_last_logits(logits))[:, None]
    streamer(_rows(token))                                      # D2H copy = the per-token sync the reference has (mx.eval)
    prompt_time = tic()
    greedy_loop(model, token, cache, max_tokens - 1, streamer, token_stopper, logit_stopper, mask, pids)
    result, gen_len = streamer.end()
    gen_time = tic()
    prompt_len = dict_input["input_ids"].size
    prompt_tps = prompt_len / prompt_time
    gen_tps = (gen_len - 1) / gen_time
    if verbose:
        print(f"\nPrompt: {prompt_tps:.2f} tokens-per-sec ({prompt_len} tokens / {prompt_time:.1f} sec)")
        print(f"Generate: {gen_tps:.2f} tokens-per-sec ({gen_len} tokens / {gen_time:.1f} sec)")
    if return_tps:
        return prompt_tps, gen_tps
    return result


def generate(prompt, images=None, preload=None, blind_model=False, quantize_model=False, quantize_cache=False,
             use_adapter=False, max_tokens=512, verbose=True, return_tps=False, early_stop=False, stream=True,
             apply_chat_template=True, enable_api=False):
    """reference phi_3_vision_mlx.py:1324-1374."""
    if "<|api_input|>" in prompt and enable_api:
        raise NotImplementedError("the <|api_input|> tool hook is outside the inference hot path of this build")
    if preload is None:
        preload = load(blind_model=blind_model, quantize_model=quantize_model, quantize_cache=quantize_cache, use_adapter=use_adapter)
    return _generate(*preload, *_apply_chat_template(prompt, images, verbose, apply_chat_template), max_tokens=max_tokens,
                     verbose=verbose, return_tps=return_tps, early_stop=early_stop, stream=stream)


# ----------------------------------------------------------------------------- choose
def _choose_from(model, processor, prompt, choices="ABCDE", mute=False):
    """Option scoring with one prefill (reference phi_3_vision_mlx.py:466-487)."""
    def _ord(s):
        return processor([f" {i}" for i in s])["input_ids"][:, -1]
    _was_prompt_str = isinstance(prompt, str)
    options = torch.as_tensor(np.asarray(_ord(choices)), dtype=torch.long)
    dict_input = processor(prompt)
    logits, _ = model(**dict_input, max_tokens=0)
    logp = model_ops.log_softmax(_last_logits(logits))
    picked = logp[:, options.to(logp.device)].float().cpu()
    indices = torch.argmax(picked, dim=-1).tolist()              # first maximum, 5-wide host argmax
    output = [choices[i] for i in indices]
    if not mute:
        if _was_prompt_str:
            print(output[0])
        else:
            for i, o in enumerate(output):
                print(f"\n< Chosen option for prompt #{i} >\n{o}")
    if _was_prompt_str:
        output = output[0]
    return output


def choose(prompt, choices="ABCDE", images=None, preload=None, blind_model=False, quantize_model=False,
           quantize_cache=False, use_adapter=False, verbose=True, apply_chat_template=True):
    """reference phi_3_vision_mlx.py:1376-1423."""
    if preload is None:
        preload = load(blind_model=blind_model, quantize_model=quantize_model, quantize_cache=quantize_cache, use_adapter=use_adapter)
    if apply_chat_template:
        prompt, _ = _apply_chat_template(prompt, images, verbose)
    return _choose_from(*preload, prompt=prompt, choices=choices)


# ----------------------------------------------------------------------------- constrain
def _preprocess(s):
    """reference phi_3_vision_mlx.py:489-493."""
    for i in ["<|system|>", "<|user|>", "<|end|>"]:
        s = s.replace(f"{i} ", f"{i}\n").replace(f"{i}\n\n", f"{i}\n")
    s = s.replace("<|end|><|assistant|>", "<|end|>\n<|assistant|>")
    return s


BF16 = torch.bfloat16


def _sum_last(x):
    return x.float().sum(-1).to(x.dtype)


def _div(x, n):
    return (x.float() / n).to(x.dtype)


def _mean_last(x):
    """`x.mean(axis=-1)` as MLX composes it: sum (rounded to x.dtype) times 1/n (rounded to x.dtype), rounded once more
    (reference phi_3_vision_mlx.py:513 on bf16 scores; tests/golden/mlx_shim.py `mean`)."""
    return _sum_last(x) * torch.tensor(1.0 / x.shape[-1], dtype=x.dtype)


def _already(a2, a1):
    """reference phi_3_vision_mlx.py:495-498: 1 where the row does NOT yet end with a1."""
    if a2.shape[1] < a1.shape[0]:
        return torch.ones(a2.shape[0])
    return (~torch.all(a2[:, -len(a1):] == a1, dim=1)).float()


def constrain_tokens(model, dict_input, constraint, id_constraint, use_beam=False, log_norm=False, trace=None):
    """One (max_new, text) constraint of the reference's `_constrain`
    (phi_3_vision_mlx.py:537-601).  Vocabulary-wide work (log-softmax, argmax,
    top-3) runs in HIP kernels; the [B, C]-sized score bookkeeping is host-side
    on CPU tensors in the logits dtype, like the reference keeps it in bf16.
    Returns (synth_sofar [B, *] int64 padded with ID_EOS, score_sofar [B]).
    trace: optional list receiving (kind, outcome ids / booleans) per data-dependent decision (parity tests walk it
    against the oracle's trace)."""
    dev = model.device
    ar = torch.arange

    def _note(kind, outcome):
        if trace is not None:
            trace.append((kind, torch.as_tensor(outcome).reshape(-1).tolist()))

    def _log_mean(x):
        return _div(_sum_last(x), math.log(x.shape[-1]) if log_norm else x.shape[-1])

    def lsm(logits):                                          # [B, L, V] -> log-probs, same shape (device)
        return model_ops.log_softmax(logits.contiguous())

    def pick(lp, pos_idx, tok):                               # lp[b, pos_idx[j], tok[b, j]] -> host [B, J]
        tok = torch.as_tensor(tok).to(dev).long()
        return lp[ar(lp.shape[0], device=dev)[:, None], torch.as_tensor(pos_idx).to(dev)[None, :], tok].cpu()

    def amax(lp, pos):                                        # host int64 [B]
        return model_ops.argmax(lp[:, pos, :].contiguous()).long().cpu()

    idc = torch.as_tensor(np.asarray(id_constraint)).long()
    C_ = idc.shape[0]
    Bn = np.asarray(dict_input["input_ids"]).shape[0]
    synth_pad = torch.full((Bn, 1), ID_EOS, dtype=torch.long)
    tiled = idc[None].expand(Bn, -1)

    def _get_beam(lp, cache, beam_idx=0, n_beam=3):
        token = amax(lp, beam_idx)
        arg_beam = model_ops.topk(lp[:, beam_idx, :].contiguous(), n_beam).long().cpu()      # (-value, index) order, Q9
        _note("argmax", token), _note("top3_set", arg_beam.sort(dim=-1).values)
        beam = torch.cat([arg_beam.reshape(-1)[:, None], idc[None].expand(Bn * n_beam, -1)], dim=-1)
        bl, _ = model(input_ids=beam, cache=cache, n_beam=n_beam, advance_offset=0)
        bl = lsm(bl)
        s0 = lp[ar(Bn, device=dev)[:, None], beam_idx, arg_beam.to(dev)].cpu().reshape(-1)[:, None]
        s1 = pick(bl, ar(C_), beam[:, 1:])
        score_all = torch.cat([s0, s1], dim=1)
        mean = _mean_last(score_all)
        am = torch.argmax(mean.reshape(-1, n_beam).float(), dim=-1)
        _note("beam_pick", arg_beam[ar(Bn), am])
        return token, arg_beam[ar(Bn), am], score_all.reshape(Bn, n_beam, -1)[ar(Bn), am]

    logits, cache = model(**dict_input, max_tokens=constraint[0] + C_ + 10)
    lp = lsm(logits)                                          # [B, 1, V] (last prompt position)
    score_0 = pick(lp, torch.tensor([lp.shape[1] - 1]), idc[0].expand(Bn)[:, None])[:, 0]
    lr, _ = model(input_ids=tiled, cache=cache, advance_offset=0)
    lr = lsm(lr)
    score_1 = pick(lr, ar(C_ - 1), tiled[:, 1:])
    running_score = lp[:, -1, :].float().max(dim=-1).values.to(lp.dtype).cpu()[:, None]
    pre_score = _log_mean(torch.cat([score_0[:, None], score_1], dim=1))
    pre_synth = torch.cat([tiled, synth_pad], dim=1)
    if use_beam and constraint[0] > 0:
        token, beam_token, beam_score = _get_beam(lp, cache, -1)
        post_score = _log_mean(beam_score)
        post_synth = torch.cat([beam_token[:, None], tiled], dim=1)
        win = pre_score > post_score
        _note("pre_vs_post", win)
        score_sofar = torch.where(win, pre_score, post_score)
        synth_sofar = torch.where(win[:, None], pre_synth, post_synth)
    else:
        token = amax(lp, -1)
        _note("argmax", token)
        score_sofar, synth_sofar = pre_score, pre_synth
    token = token[:, None]
    tokens = []
    finished = torch.ones(Bn)
    for _ in range(constraint[0]):
        tokens.append(token)
        token_plus = torch.cat([token, tiled], dim=1)
        logits, cache = model(input_ids=token_plus, cache=cache, advance_offset=1)
        lp = lsm(logits)
        g = pick(lp, ar(C_), token_plus[:, 1:])
        pre_score = _log_mean(torch.cat([running_score, g], dim=1))
        pre_synth = torch.cat(tokens + [tiled, synth_pad], dim=1)
        if use_beam:
            token, beam_token, beam_score = _get_beam(lp, cache)
            post_score = _log_mean(torch.cat([running_score, beam_score], dim=1))
            post_synth = torch.cat(tokens + [beam_token[:, None], tiled], dim=1)
            win = pre_score > post_score
            _note("pre_vs_post", win)
            score = torch.where(win, pre_score, post_score)
            synth = torch.where(win[:, None], pre_synth, post_synth)
        else:
            token = amax(lp, 0)
            _note("argmax", token)
            score, synth = pre_score, pre_synth
        synth_sofar = torch.cat([synth_sofar, synth_pad], dim=1)
        finished = finished * _already(torch.cat(tokens, dim=1), idc)
        upd = (score > score_sofar).float() * finished
        _note("update", upd)
        synth_sofar = torch.where(upd[:, None] > 0, synth, synth_sofar)
        score_sofar = torch.where(upd > 0, score, score_sofar)
        running_score = torch.cat([running_score, pick(lp, torch.tensor([0]), token[:, None])], dim=1)
        finished = finished * (token != ID_EOS).float()
        if finished.sum() < 1:
            break
        token = token[:, None]
    return synth_sofar, score_sofar


def _constrain(model, processor, prompt, constraints, return_full_text=False, mute=False, use_beam=False, verbose=True,
               log_norm=False):
    """reference phi_3_vision_mlx.py:500-619."""
    _was_prompt_str = isinstance(prompt, str)
    if _was_prompt_str:
        prompt = [prompt]
    tic = Tic()
    constrain_time = 0
    prompt = [_preprocess(s) for s in prompt]
    len_ps = [len(p) for p in prompt]
    output = prompt
    for constraint in constraints:
        if isinstance(constraint, str):
            _output = _choose_from(model, processor, prompt, constraint, True)
            output = [" ".join([p, o]) for p, o in zip(prompt, _output)]
            prompt = output
            continue
        id_constraint = processor.tokenizer.encode(constraint[1], add_special_tokens=False)[1:]
        dict_input = processor(prompt)
        synth_sofar, _ = constrain_tokens(model, dict_input, constraint, id_constraint, use_beam, log_norm)
        constrain_time += tic()
        ids = np.asarray(dict_input["input_ids"])
        output = np.concatenate([ids, synth_sofar.numpy()], axis=1).tolist()
        S = ids.shape[1]
        output = [(i[:i.index(ID_EOS, S)] if ID_EOS in i[S:] else i) for i in output]
        output = [[num for num in sublist if num not in (0, 1)] for sublist in output]
        output = processor.tokenizer.batch_decode(output)
        output = [_preprocess(s) for s in output]
        prompt = output
    if not return_full_text:
        output = [o[l:] for o, l in zip(output, len_ps)]
    if not mute:
        if _was_prompt_str:
            print(output[0])
        else:
            for i, o in enumerate(output):
                print(f"\n< Constrained text for prompt #{i} >\n{o}")
    if verbose:
        print(f"Constrain: {constrain_time:.2f} sec")
    if _was_prompt_str:
        output = output[0]
    return output


def constrain(prompt, constraints=[(0, "\nThe"), (100, " The correct answer is"), "ABCDE"], images=None, preload=None,
              blind_model=False, quantize_model=False, quantize_cache=False, use_adapter=False, verbose=True,
              apply_chat_template=True, use_beam=False):
    """reference phi_3_vision_mlx.py:1425-1487."""
    if preload is None:
        preload = load(blind_model=blind_model, quantize_model=quantize_model, quantize_cache=quantize_cache, use_adapter=use_adapter)
    if apply_chat_template:
        prompt = _apply_chat_template(prompt, None, verbose)[0]
    return _constrain(*preload, prompt=prompt, constraints=constraints, use_beam=use_beam, verbose=verbose)


# ----------------------------------------------------------------------------- benchmark
BENCHMARK_IMAGE_URL = "https://collectionapi.metmuseum.org/api/collection/v1/iiif/344291/725918/main-image"
BENCHMARK_BATCH = [                                            # the reference's list (:1226-1243): 16 literals, 15 prompts --
    "Write an executive summary for a communications business plan",     # a missing comma fuses two of them (kept: same workload)
    "Explain quantum computing.",
    "Write a poem about the first snowfall of the year.",
    "Write a Python function to implement a neural network from scratch, with detailed comments.",
    "Write a resume.",
    "Explain the key concepts of quantum computing and provide a Rust code example demonstrating quantum superposition.",
    "Explain the concept of dark matter and its significance in the universe.",
    "Summarize the major events of the French Revolution.",
    "Describe the water cycle.",
    "Write a Neurology ICU Admission Note.",
    "Describe a bustling alien marketplace on a distant planet with unique goods and creatures."
    "Imagine you have a magic potion that grants one wish. What would you wish for and how would it change your life?",
    "Compose a limerick about a clumsy robot.",
    "Write a JavaScript function to sort an array of objects by a specific property.",
    "Design a database schema for a social media platform, considering user profiles, posts, and interactions.",
    "Implement a basic encryption algorithm in Python.",
]


def _format_benchmark(json_path="benchmark.json"):
    """reference phi_3_vision_mlx.py:427-443: the README's table (decode tokens-per-second per task and variant)."""
    with open(json_path, "r") as f:
        data = json.load(f)
    names = ["Text Generation", "Image Captioning", "Batched Generation"]
    table = """
    | Task                  | Vanilla Model | Quantized Model | Quantized Cache | LoRA Adapter |
    |-----------------------|---------------|-----------------|-----------------|--------------|"""
    for i, name in enumerate(names):
        if i >= len(data["vanilla"]):
            break
        v, qm, qc, lo = (data[k][i][2] for k in ("vanilla", "q_model", "q_cache", "lora"))
        table += f"\n    | {name}{' ' * (22 - len(name))}|  {v:.2f} tps     |  {qm:.2f} tps      |  {qc:.2f} tps       |  {lo:.2f} tps    |"
    print(table)
    return table


def _benchmark_adapter(model, path):
    """The LoRA column needs an adapter; the reference trains one first (`train_lora(take=1)`, :1246-1252 -- training is
    outside this build).  When `path` holds none, write a seeded synthetic one of the shape `train_lora` produces by
    default (last layer, qkv_proj, rank 1, alpha = rank, scale 1; :898, :1011): the kernels do the same work per token."""
    from .weights import ADAPTER_CONFIG, save_adapter
    if os.path.exists(os.path.join(path, ADAPTER_CONFIG)):
        return path
    cfg = model.cfg
    H, hd = cfg.hidden_size, cfg.hidden_size // cfg.num_attention_heads
    n_qkv = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * hd
    gen = torch.Generator().manual_seed(0)
    i = cfg.num_hidden_layers - 1
    tensors = {f"model.layers.{i}.self_attn.qkv_proj.lora_a": (torch.rand((H, 1), generator=gen) * 2 - 1) * H ** -0.5,
               f"model.layers.{i}.self_attn.qkv_proj.lora_b": torch.randn((1, n_qkv), generator=gen) * 0.01}
    save_adapter(path, {"model_path": "synthetic", "adapter_path": path, "lora_layers": 1, "lora_targets": ["self_attn.qkv_proj"],
                        "lora_parameters": {"rank": 1, "alpha": 1, "dropout": 0.0, "scale": 1.0}}, tensors)
    return path


def benchmark(blind_model=False, json_path="benchmark.json", *, synthetic=None, image=None, max_tokens=100, adapter_path=None):
    """reference phi_3_vision_mlx.py:1178-1277, like for like: the same three tasks (text generation, image captioning,
    the 15-prompt batch), the same four variants (`vanilla`, `q_model` = quantize_model, `q_cache` = quantize_cache,
    `lora` = use_adapter), 100 new tokens, results[variant] = [[task, prompt_tps, gen_tps], ...] written to `json_path`
    and printed as the README's table (README.md:274-280 holds the reference's Apple M1 Max numbers).
    Keyword-only extras (no network / no checkpoints here): `synthetic=True` uses seeded random weights, `image` replaces
    the museum URL (a seeded 336x336 noise image when the URL cannot be fetched), `adapter_path` names an existing adapter.
    The image task is skipped for `blind_model=True` exactly as the reference's processor would warn and ignore it."""
    if image is None:
        try:
            image = _load_image(BENCHMARK_IMAGE_URL)
        except Exception:
            from PIL import Image
            image = Image.fromarray(np.random.default_rng(0).integers(0, 256, (336, 336, 3), dtype=np.uint8))
    prompts = [("Write a mystery horror.",), ("What is shown in this image?", image), (list(BENCHMARK_BATCH), None)]
    results = {"vanilla": [], "q_model": [], "q_cache": [], "lora": []}
    for method in results:
        kwargs = {"blind_model": blind_model}
        if synthetic:
            kwargs["synthetic"] = synthetic
        if method == "q_model":
            kwargs["quantize_model"] = True
        elif method == "q_cache":
            kwargs["quantize_cache"] = True
        preload = load(**kwargs)
        if method == "lora":
            ap = adapter_path or _benchmark_adapter(preload[0], os.path.join(PATH_ADAPTERS, "benchmark_synthetic" + ("_blind" if blind_model else "")))
            _attach_adapter(preload[0], ap)
        for i, prompt in enumerate(prompts):
            if blind_model and i == 1:
                prompt = prompt[:1]                                # text-only model: the image is ignored (phi.py:247-249)
            prompt_tps, gen_tps = generate(*prompt, preload=preload, max_tokens=max_tokens, return_tps=True, verbose=False)
            results[method].append([i, prompt_tps, gen_tps])
        del preload
        torch.cuda.empty_cache()
    with open(json_path, "w") as f:
        json.dump(results, f, indent=4)
    _format_benchmark(json_path)
    return results
