"""A FUNCTIONAL stand-in for `mlx.core` / `mlx.nn` / `mlx.utils` over torch-CPU (test infrastructure, build container only).

Purpose: `mlx==0.15.0` (reference requirements.txt:1) cannot be installed here, so the reference's model code
(`/root/reference/phi.py:135-617`) and decoding loops (`phi_3_vision_mlx.py:376-409,466-619`) never ran anywhere in this
pipeline.  With this module installed into `sys.modules`, `import phi` / `import phi_3_vision_mlx` succeed and the
REFERENCE'S OWN composition -- its reshapes, transposes, slice assignments, cache/rewind/beam logic, mask and RoPE
construction, HD-merge order, loop control flow -- executes on CPU.  `gen_golden_refmodel.py` uses it to write
`ref_model_tiny.npz`; the oracle (`oracle/phi3v_oracle.py`) and the HIP path are then pinned to those outputs.

What this pins and what it cannot.  It pins everything the reference's Python decides.  It cannot pin the bits of MLX's
Metal kernels: each primitive here follows MLX's DOCUMENTED semantics and nothing finer --
  * dtype promotion: bf16 (x) fp32 -> fp32; python scalars are weak; ints (x) floats -> float (torch's rules coincide);
  * every primitive rounds its result ONCE to its output dtype; matmul / reductions accumulate in fp32;
  * `nn.Linear` = x @ W.T (+ b), `nn.Embedding` = W[ids] (negative ids wrap, numpy-style);
  * `nn.RMSNorm` = `mx.fast.rms_norm`: fp32 statistics, `w * astype(x * rsqrt(mean(x^2) + eps), x.dtype)` -- the normalised
    row is rounded to the activation dtype BEFORE the weight multiply (two roundings for bf16 x);
  * `nn.LayerNorm` (eps 1e-5, biased variance) in fp32 for fp32 activations;
  * `nn.gelu_fast_approx(x) = x * sigmoid(1.702 x)`; `nn.GELU()` = exact erf; `nn.silu(x) = x * sigmoid(x)` (two roundings);
  * `nn.log_softmax(x) = x - logsumexp(x)` -- the composite MLX defines: logsumexp is rounded to x.dtype, then subtracted;
  * `mx.mean(x) = sum(x) * array(1/n, dtype)` -- MLX's composite: the sum and the reciprocal count are both rounded to x.dtype;
  * `mx.softmax`, `mx.fast.scaled_dot_product_attention` = softmax(scale * q k^T [+ mask]) v with fp32 internals;
  * `mx.argmax` returns the FIRST maximum; `mx.repeat` = repeat-interleave; `mx.tile`/`mx.split`/`mx.pad` numpy semantics;
  * slice assignment casts to the destination dtype and drops leading unit axes of the update.
Two places where MLX's result is implementation-defined get a deterministic definition:
  Q7  `softmax` of a row that is -inf everywhere (left-pad QUERY rows under Mask4D, phi.py:553-559) is NaN in IEEE
      arithmetic and undefined under Metal fast-math.  Here such a row is all zeros.  Valid rows do not depend on the
      choice: pad KEYS always carry -inf, so any FINITE pad-row output gives them exactly zero weight downstream.
  Q9  `mx.argpartition(-x, kth)[:, :kth]` returns the kth smallest in unspecified order; here ordered by (value, index).
mx.quantize / dequantize / quantized_matmul and nn.quantize (QuantizedLinear, QuantizedEmbedding) follow MLX's affine group
format as `weights.mlx_quantize` restates it.  Not implemented (raise): optimizers, value_and_grad -- off the pinned path.
"""
import builtins as _b
import importlib.machinery
import math
import sys
import types

import numpy as np
import torch

torch.set_grad_enabled(False)


# ---------------------------------------------------------------------------------------------------------------------
# dtypes
# ---------------------------------------------------------------------------------------------------------------------
class Dtype:
    def __init__(self, name, t):
        self.name, self.t = name, t

    def __repr__(self):
        return f"mlx.core.{self.name}"

    def __eq__(self, o):
        return isinstance(o, Dtype) and o.t == self.t

    def __hash__(self):
        return hash(self.t)


float32, float16, bfloat16 = Dtype("float32", torch.float32), Dtype("float16", torch.float16), Dtype("bfloat16", torch.bfloat16)
int8, int16, int32, int64 = (Dtype(n, t) for n, t in (("int8", torch.int8), ("int16", torch.int16), ("int32", torch.int32), ("int64", torch.int64)))
uint8, uint32 = Dtype("uint8", torch.uint8), Dtype("uint32", torch.int64)      # torch-CPU has no useful uint32: carried as int64
bool_ = Dtype("bool", torch.bool)
_BY_TORCH = {d.t: d for d in (float32, float16, bfloat16, int8, int16, int32, int64, uint8, bool_)}
inf = float("inf")
nan = float("nan")
pi = math.pi


def _td(dtype):
    return None if dtype is None else dtype.t


def _t(x):
    """shim array / python / numpy -> torch tensor or python scalar (scalars stay weak)."""
    if isinstance(x, array):
        return x._t
    if isinstance(x, (bool, int, float)):
        return x
    if isinstance(x, np.generic):
        return x.item()
    return _from_python(x)


def _from_python(x, dtype=None):
    """mx.array(...) conversion rules: python floats / numpy float64 -> float32; python ints -> int32; bools -> bool."""
    if isinstance(x, array):
        t = x._t
    elif isinstance(x, torch.Tensor):
        t = x
    elif isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        if t.dtype == torch.float64:
            t = t.to(torch.float32)
    else:
        def _has_array(v):
            return isinstance(v, array) or (isinstance(v, (list, tuple)) and _b.any(_has_array(u) for u in v))
        if _has_array(x):
            def _plain(v):
                return v.tolist() if isinstance(v, array) else [_plain(u) for u in v] if isinstance(v, (list, tuple)) else v
            x = _plain(x)
        a = np.asarray(x)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.int64:
            a = a.astype(np.int32)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype.t)
    return t


def _axes(axis, ndim):
    if axis is None:
        return tuple(range(ndim))
    if isinstance(axis, int):
        return (axis,)
    return tuple(axis)


def _index(idx):
    """Translate an MLX index expression to torch: shim arrays -> long tensors (bool stays bool)."""
    def one(i):
        if isinstance(i, array):
            return i._t if i._t.dtype == torch.bool else i._t.long()
        if isinstance(i, (list, np.ndarray)):
            return torch.as_tensor(np.asarray(i)).long()
        return i
    if isinstance(idx, tuple):
        return tuple(one(i) for i in idx)
    return one(idx)


class array:
    """mlx.core.array over a torch CPU tensor."""
    __array_priority__ = 1000

    def __init__(self, val, dtype=None):
        self._t = _from_python(val, dtype)

    # -- introspection -------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return _BY_TORCH[self._t.dtype]

    @property
    def size(self):
        return self._t.numel()

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def T(self):
        return self.transpose()

    def __len__(self):
        return self._t.shape[0]

    def __iter__(self):
        return (array(self._t[i]) for i in range(self._t.shape[0]))

    def __bool__(self):
        return bool(self._t)

    def __float__(self):
        return float(self._t)

    def __int__(self):
        return int(self._t)

    def __index__(self):
        return int(self._t)

    def __repr__(self):
        return f"array({self._t.tolist() if self._t.numel() < 64 else self.shape}, dtype={self.dtype})"

    def __array__(self, dtype=None, copy=None):
        t = self._t.float() if self._t.dtype == torch.bfloat16 else self._t
        a = t.numpy()
        return a.astype(dtype) if dtype is not None else a

    def item(self):
        return self._t.item()

    def tolist(self):
        return (self._t.float() if self._t.dtype == torch.bfloat16 else self._t).tolist()

    # -- shape ops -----------------------------------------------------------------------------------------------
    def astype(self, dtype):
        return array(self._t.to(dtype.t))

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return array(self._t.reshape(shape))

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes:
            axes = tuple(reversed(range(self._t.dim())))
        return array(self._t.permute(*axes))

    def squeeze(self, axis=None):
        return squeeze(self, axis)

    def flatten(self, start_axis=0, end_axis=-1):
        return flatten(self, start_axis, end_axis)

    def swapaxes(self, a, b):
        return array(self._t.transpose(a, b))

    # -- reductions / unary ----------------------------------------------------------------------------------------
    def sum(self, axis=None, keepdims=False):
        return sum(self, axis, keepdims)

    def mean(self, axis=None, keepdims=False):
        return mean(self, axis, keepdims)

    def max(self, axis=None, keepdims=False):
        return max(self, axis, keepdims)

    def min(self, axis=None, keepdims=False):
        return min(self, axis, keepdims)

    def all(self, axis=None, keepdims=False):
        return all(self, axis, keepdims)

    def any(self, axis=None, keepdims=False):
        return any(self, axis, keepdims)

    def square(self):
        return self * self

    def abs(self):
        return array(self._t.abs())

    def exp(self):
        return exp(self)

    def log(self):
        return log(self)

    def sqrt(self):
        return array(torch.sqrt(self._t))

    def rsqrt(self):
        return array(torch.rsqrt(self._t))

    # -- indexing ------------------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        return array(self._t[_index(idx)])

    def __setitem__(self, idx, val):
        dst = self._t[_index(idx)]
        v = _t(val)
        if isinstance(v, torch.Tensor):
            while v.dim() > dst.dim() and v.shape[0] == 1:       # MLX drops leading unit axes of the update
                v = v[0]
            v = v.to(self._t.dtype)
        self._t[_index(idx)] = v

    # -- arithmetic (torch's promotion rules == MLX's for every combination the reference uses) ---------------------
    def _bin(self, o, f, rev=False):
        a, b = self._t, _t(o)
        return array(f(b, a) if rev else f(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)
    def __rtruediv__(self, o): return self._bin(o, torch.true_divide, True)
    def __floordiv__(self, o): return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode="floor"))
    def __mod__(self, o): return self._bin(o, torch.remainder)
    def __pow__(self, o): return self._bin(o, torch.pow)

    def __rpow__(self, o):
        b = self._t
        return array(torch.pow(torch.tensor(o, dtype=b.dtype) if b.is_floating_point() else torch.tensor(o), b))

    def __neg__(self): return array(-self._t)
    def __invert__(self): return array(~self._t)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __eq__(self, o): return self._bin(o, torch.eq)
    def __ne__(self, o): return self._bin(o, torch.ne)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __and__(self, o): return self._bin(o, torch.bitwise_and)
    def __or__(self, o): return self._bin(o, torch.bitwise_or)
    __hash__ = None

    def __iadd__(self, o):
        self._t = torch.add(self._t, _t(o))
        return self

    def __isub__(self, o):
        self._t = torch.sub(self._t, _t(o))
        return self

    def __imul__(self, o):
        """`a *= b`: MLX rebinds to the promoted product (e.g. float32 ones *= bool mask -> float32)."""
        self._t = torch.mul(self._t, _t(o))
        return self

    def __itruediv__(self, o):
        self._t = torch.true_divide(self._t, _t(o))
        return self


# ---------------------------------------------------------------------------------------------------------------------
# mlx.core functions
# ---------------------------------------------------------------------------------------------------------------------
def _arr(x):
    return x if isinstance(x, array) else array(x)


def eval(*args, **kwargs):      # noqa: A001 -- lazy evaluation barrier: nothing to do on an eager backend
    return None


def compile(fun=None, inputs=None, outputs=None, shapeless=False):      # noqa: A001
    if fun is None:
        return lambda f: f
    return fun


def zeros(shape, dtype=float32):
    return array(torch.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=dtype.t))


def ones(shape, dtype=float32):
    return array(torch.ones(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=dtype.t))


def full(shape, vals, dtype=None):
    t = torch.full(tuple(shape) if not isinstance(shape, int) else (shape,), _t(vals) if not isinstance(vals, array) else vals.item())
    if t.dtype == torch.int64:
        t = t.to(torch.int32)
    return array(t, dtype)


def zeros_like(x):
    return array(torch.zeros_like(x._t))


def ones_like(x):
    return array(torch.ones_like(x._t))


def arange(*args, dtype=None, **kw):
    """mx.arange(stop) / (start, stop[, step]); int arguments -> int32, float -> float32 unless dtype is given."""
    t = torch.arange(*args)
    if dtype is None:
        dtype = float32 if t.is_floating_point() else int32
    return array(torch.arange(*args, dtype=dtype.t))


def linspace(start, stop, num=50, dtype=float32):
    return array(torch.linspace(start, stop, num, dtype=dtype.t))


def matmul(a, b):
    """fp32 accumulation, one rounding to the promoted dtype."""
    a, b = _arr(a)._t, _arr(b)._t
    out = torch.promote_types(a.dtype, b.dtype)
    return array(torch.matmul(a.to(torch.float32), b.to(torch.float32)).to(out))


def addmm(c, a, b, alpha=1.0, beta=1.0):
    a, b, c = _arr(a)._t, _arr(b)._t, _arr(c)._t
    out = torch.promote_types(torch.promote_types(a.dtype, b.dtype), c.dtype)
    return array((alpha * torch.matmul(a.float(), b.float()) + beta * c.float()).to(out))


def concatenate(arrays, axis=0):
    return array(torch.cat([_arr(a)._t for a in arrays], dim=axis))


def stack(arrays, axis=0):
    return array(torch.stack([_arr(a)._t for a in arrays], dim=axis))


def split(a, indices_or_sections, axis=0):
    s = indices_or_sections
    parts = torch.tensor_split(a._t, s if isinstance(s, int) else list(s), dim=axis)
    return [array(p) for p in parts]


def repeat(a, repeats, axis=None):
    a = _arr(a)
    if axis is None:
        return array(a._t.reshape(-1).repeat_interleave(repeats))
    return array(a._t.repeat_interleave(repeats, dim=axis))


def tile(a, reps):
    return array(torch.tile(_arr(a)._t, (reps,) if isinstance(reps, int) else tuple(reps)))


def broadcast_to(a, shape):
    return array(torch.broadcast_to(_arr(a)._t, tuple(shape)))


def expand_dims(a, axis):
    t = a._t
    for ax in sorted(_axes(axis, t.dim() + (1 if isinstance(axis, int) else len(axis)))):
        t = t.unsqueeze(ax)
    return array(t)


def squeeze(a, axis=None):
    t = a._t
    if axis is None:
        return array(t.squeeze())
    for ax in sorted((x % t.dim() for x in _axes(axis, t.dim())), reverse=True):
        t = t.squeeze(ax)
    return array(t)


def flatten(a, start_axis=0, end_axis=-1):
    return array(torch.flatten(a._t, start_axis, end_axis))


def reshape(a, shape):
    return a.reshape(shape)


def transpose(a, axes=None):
    return a.transpose(*(axes or ()))


def swapaxes(a, x, y):
    return array(a._t.transpose(x, y))


def pad(a, pad_width, constant_values=0):
    """numpy-style pad_width: int | (before, after) | ((b0, a0), (b1, a1), ...)."""
    t = a._t
    if isinstance(pad_width, int):
        pad_width = [(pad_width, pad_width)] * t.dim()
    elif isinstance(pad_width[0], int):
        pad_width = [tuple(pad_width)] * t.dim()
    flat = []
    for b, e in reversed(list(pad_width)):
        flat += [b, e]
    return array(torch.nn.functional.pad(t, flat, value=constant_values))


def triu(a, k=0):
    return array(torch.triu(a._t, diagonal=k))


def tril(a, k=0):
    return array(torch.tril(a._t, diagonal=k))


def where(cond, x, y):
    c, xv, yv = _t(cond), _t(x), _t(y)
    if not isinstance(c, torch.Tensor):
        c = torch.tensor(c)
    c = c if c.dtype == torch.bool else c != 0
    if not isinstance(xv, torch.Tensor) and not isinstance(yv, torch.Tensor):
        flt = isinstance(xv, float) or isinstance(yv, float)
        xv, yv = (torch.tensor(v, dtype=torch.float32 if flt else torch.int32) for v in (xv, yv))
    return array(torch.where(c, xv, yv))


def _reduce(f):
    def g(a, axis=None, keepdims=False):
        t = _arr(a)._t
        ax = _axes(axis, t.dim())
        return array(f(t, ax, keepdims))
    return g


sum = _reduce(lambda t, ax, kd: t.to(torch.float32).sum(dim=ax, keepdim=kd).to(t.dtype) if t.is_floating_point()      # noqa: A001
              else t.sum(dim=ax, keepdim=kd).to(torch.int32 if t.dtype in (torch.bool, torch.int32) else t.dtype))


def mean(a, axis=None, keepdims=False):
    """MLX's composite (ops.cpp): sum(a) * array(1 / n, dtype) -- the sum is rounded to the dtype, and so is 1 / n."""
    t = _arr(a)._t
    n = 1
    for ax in _axes(axis, t.dim()):
        n *= t.shape[ax]
    s = sum(a, axis, keepdims)._t
    if not s.is_floating_point():
        s = s.to(torch.float32)
    return array(s * torch.tensor(1.0 / n, dtype=s.dtype))


max = _reduce(lambda t, ax, kd: torch.amax(t, dim=ax, keepdim=kd))      # noqa: A001
min = _reduce(lambda t, ax, kd: torch.amin(t, dim=ax, keepdim=kd))      # noqa: A001
all = _reduce(lambda t, ax, kd: torch.all(t != 0 if t.dtype != torch.bool else t, dim=ax, keepdim=kd) if len(ax) == 1      # noqa: A001
              else torch.all(t != 0 if t.dtype != torch.bool else t).reshape([1] * t.dim() if kd else []))
any = _reduce(lambda t, ax, kd: torch.any(t != 0 if t.dtype != torch.bool else t, dim=ax, keepdim=kd) if len(ax) == 1      # noqa: A001
              else torch.any(t != 0 if t.dtype != torch.bool else t).reshape([1] * t.dim() if kd else []))


def argmax(a, axis=None, keepdims=False):
    """First maximum.  Computed on fp32 copies of bf16 values (exact), so ties resolve by index, never by rounding."""
    t = a._t.to(torch.float32) if a._t.is_floating_point() else a._t
    if axis is None:
        return array(torch.argmax(t.reshape(-1)).to(torch.int32))
    return array(torch.argmax(t, dim=axis, keepdim=keepdims).to(torch.int32))


def argmin(a, axis=None, keepdims=False):
    return argmax(-a, axis, keepdims)


def argsort(a, axis=-1):
    t = a._t.to(torch.float32) if a._t.is_floating_point() else a._t
    return array(torch.sort(t, dim=axis, stable=True).indices.to(torch.int32))


def argpartition(a, kth, axis=-1):
    """Q9: MLX leaves the order inside each side of the partition unspecified; a stable full argsort is one valid answer."""
    return argsort(a, axis)


def sort(a, axis=-1):
    return array(torch.sort(a._t, dim=axis, stable=True).values)


def _unary(f):
    def g(a):
        t = _arr(a)._t
        if not t.is_floating_point():
            t = t.to(torch.float32)
        return array(f(t.to(torch.float32)).to(t.dtype))
    return g


exp, log, cos, sin, sqrt, rsqrt, tanh, erf = (_unary(f) for f in (torch.exp, torch.log, torch.cos, torch.sin, torch.sqrt, torch.rsqrt,
                                                                   torch.tanh, torch.erf))
sigmoid = _unary(torch.sigmoid)


def abs(a):      # noqa: A001
    return array(_arr(a)._t.abs())


def square(a):
    return _arr(a) * _arr(a)


def maximum(a, b):
    return _arr(a)._bin(b, lambda x, y: torch.maximum(x, torch.as_tensor(y, dtype=x.dtype)))


def minimum(a, b):
    return _arr(a)._bin(b, lambda x, y: torch.minimum(x, torch.as_tensor(y, dtype=x.dtype)))


def isinf(a):
    return array(torch.isinf(a._t))


def isnan(a):
    return array(torch.isnan(a._t))


def stop_gradient(a):
    return a


def logsumexp(a, axis=None, keepdims=False):
    t = a._t
    ax = _axes(axis, t.dim())
    return array(torch.logsumexp(t.to(torch.float32), dim=ax, keepdim=keepdims).to(t.dtype))


def softmax(a, axis=-1, precise=False):
    """fp32 internals, one rounding.  A row that is -inf everywhere -> zeros (Q7, see the module docstring)."""
    t = a._t
    w = t.to(torch.float32)
    m = torch.amax(w, dim=axis, keepdim=True)
    m = torch.where(torch.isinf(m) & (m < 0), torch.zeros_like(m), m)
    e = torch.exp(w - m)
    s = e.sum(dim=axis, keepdim=True)
    return array(torch.where(s > 0, e / s.clamp_min(1e-38), torch.zeros_like(e)).to(t.dtype))


def take(a, indices, axis=None):
    idx = _arr(indices)._t.long()
    if axis is None:
        return array(a._t.reshape(-1)[idx])
    return array(torch.index_select(a._t, axis, idx.reshape(-1)).reshape(a.shape[:axis] + tuple(idx.shape) + a.shape[axis + 1:]))


def load(path):
    from safetensors.torch import load_file
    return {k: array(v) for k, v in load_file(path).items()}


def save_safetensors(path, d, metadata=None):
    from safetensors.torch import save_file
    save_file({k: v._t.contiguous() for k, v in d.items()}, path if path.endswith(".safetensors") else path + ".safetensors", metadata)


def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"mlx shim: {name} is off the pinned path")
    return f


def round(a, decimals=0):      # noqa: A001
    """mx.round: to the nearest integer, halves to even (the backends' `rint`), in the array's dtype."""
    t = _arr(a)._t
    if decimals != 0:
        raise NotImplementedError("mlx shim: round(decimals != 0) is off the pinned path")
    return array(torch.round(t.to(torch.float32)).to(t.dtype) if t.is_floating_point() else t)


def clip(a, a_min=None, a_max=None):
    """mx.clip = minimum(maximum(a, a_min), a_max)."""
    out = _arr(a)
    if a_min is not None:
        out = maximum(out, a_min)
    if a_max is not None:
        out = minimum(out, a_max)
    return out


# The affine group quantiser.  The reference pins mlx==0.15.0 (requirements.txt:1), where `mx.quantize` / `mx.dequantize` are not
# kernels but COMPOSITES of array primitives (mlx/ops.cpp; the fused affine_quantize kernel arrived with 0.17), restated here
# primitive by primitive over this file's own array ops -- so every intermediate is an array of w's dtype and rounds to it, exactly as
# the composite's intermediates do.  Nothing of the product is imported: `phi_3_vision_mlx_amd.weights.mlx_quantize` is a SECOND
# statement of the same algorithm and tests/test_host_logic.py cross-checks the two bit for bit on random inputs.
def quantize(w, group_size=64, bits=4):
    """w [N, K] -> (uint32 words [N, K * bits / 32] carried as int32 bit patterns, scales [N, K / group], biases [N, K / group]):
    per group  w ~ scale * q + bias,  q in 0 .. 2^bits - 1, the range end of larger magnitude represented exactly;
    element k of a word at bits [bits * k, bits * (k + 1))."""
    w = _arr(w)
    if w.ndim != 2 or w.shape[1] % group_size or group_size % (32 // bits):
        raise ValueError(f"[quantize] the last dimension ({w.shape}) must be divisible by the group size {group_size}")
    dt = w.dtype
    n_bins = (1 << bits) - 1
    el_per_int = 32 // bits
    shifts = array(torch.tensor([1 << sh for sh in range(0, 32, bits)], dtype=torch.int64)).reshape(1, 1, -1)    # power(2, arange(0, 32, bits))
    packed_w = reshape(w, (w.shape[0], w.shape[1] // group_size, group_size))
    w_max = max(packed_w, axis=-1, keepdims=True)
    w_min = min(packed_w, axis=-1, keepdims=True)
    mask = abs(w_min) > abs(w_max)
    scales = maximum((w_max - w_min) / array(n_bins, dt), array(1e-7, dt))
    scales = where(mask, scales, -scales)
    edge = where(mask, w_min, w_max)
    q0 = round(edge / scales)
    scales = where(q0 != array(0, dt), edge / q0, scales)
    biases = where(q0 == array(0, dt), array(0, dt), edge)
    packed_w = clip(round((packed_w - biases) / scales), array(0.0, dt), array(n_bins, dt)).astype(uint32)
    packed_w = reshape(packed_w, (w.shape[0], -1, el_per_int))
    words = sum(packed_w * shifts, axis=2)._t                              # uint32 arithmetic (carried in int64: no wrap below 2^32)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    return array(words), reshape(scales, (w.shape[0], -1)), reshape(biases, (w.shape[0], -1))


def _unpack(w, bits):
    """The composite's shift pairs: element k of every word = (word << (32 - bits * (k + 1))) >> (32 - bits), as uint32."""
    u = _arr(w)._t.to(torch.int64) & 0xFFFFFFFF
    parts = [(((u << (32 - (start + bits))) & 0xFFFFFFFF) >> (32 - bits)).unsqueeze(-1) for start in range(0, 32, bits)]
    return torch.cat(parts, dim=-1)


def dequantize(w, scales, biases, group_size=64, bits=4):
    """multiply(codes, scales) then add(., biases): two primitives, each rounding to the scales' dtype."""
    s, b = _arr(scales), _arr(biases)
    w_full = array(_unpack(w, bits).reshape(s.shape[0], -1, group_size).to(s._t.dtype))       # 0 .. 15: exact in every float dtype
    w_full = w_full * expand_dims(s, -1)
    w_full = w_full + expand_dims(b, -1)
    return reshape(w_full, (s.shape[0], -1))


def quantized_matmul(x, w, scales, biases, transpose=True, group_size=64, bits=4):
    """x @ dequantize(w).T (documented semantics of the fused kernel): scale * q + bias and the accumulation in fp32, result in
    x.dtype."""
    s, b = _arr(scales)._t.float(), _arr(biases)._t.float()
    wd = (_unpack(w, bits).reshape(s.shape[0], -1, group_size).float() * s[..., None] + b[..., None]).reshape(s.shape[0], -1)
    xt = _arr(x)._t
    return array(torch.matmul(xt.float(), wd.t() if transpose else wd).to(xt.dtype))


# ---------------------------------------------------------------------------------------------------------------------
# mlx.utils
# ---------------------------------------------------------------------------------------------------------------------
def tree_flatten(tree, prefix="", is_leaf=None):
    out = []
    if is_leaf is not None and is_leaf(tree):
        return [(prefix[1:], tree)]
    if isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            out += tree_flatten(v, f"{prefix}.{i}", is_leaf)
        return out
    if isinstance(tree, dict):
        for k, v in tree.items():
            out += tree_flatten(v, f"{prefix}.{k}", is_leaf)
        return out
    return [(prefix[1:], tree)]


def tree_unflatten(flat):
    if len(flat) == 1 and flat[0][0] == "":
        return flat[0][1]
    groups = {}
    for k, v in flat:
        head, _, rest = k.partition(".")
        groups.setdefault(head, []).append((rest, v))
    try:
        keys = sorted(groups, key=int)
        is_list = True
    except ValueError:
        keys, is_list = list(groups), False
    if is_list:
        out = []
        for k in keys:
            while len(out) < int(k):
                out.append({})
            out.append(tree_unflatten(groups[k]))
        return out
    return {k: tree_unflatten(groups[k]) for k in keys}


def tree_map(fn, tree, *rest, is_leaf=None):
    if is_leaf is not None and is_leaf(tree):
        return fn(tree, *rest)
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(fn, v, *(r[i] for r in rest), is_leaf=is_leaf) for i, v in enumerate(tree))
    if isinstance(tree, dict):
        return {k: tree_map(fn, v, *(r[k] for r in rest), is_leaf=is_leaf) for k, v in tree.items()}
    return fn(tree, *rest)


# ---------------------------------------------------------------------------------------------------------------------
# mlx.nn
# ---------------------------------------------------------------------------------------------------------------------
class Module:
    """Attribute-based stand-in for mlx.nn.Module: public attributes holding arrays / Modules / lists / dicts of them are
    the parameter tree (names starting with `_` are private state, as in MLX)."""

    def __init__(self):
        self._training = True

    def __call__(self, *a, **k):
        raise NotImplementedError

    def _public(self):
        return {k: v for k, v in vars(self).items() if not k.startswith("_")}

    @staticmethod
    def _is_param_container(v):
        if isinstance(v, (array, Module)):
            return True
        if isinstance(v, (list, tuple)):
            return builtins_any(Module._is_param_container(u) for u in v)
        if isinstance(v, dict):
            return builtins_any(Module._is_param_container(u) for u in v.values())
        return False

    def parameters(self):
        def walk(v):
            if isinstance(v, Module):
                return v.parameters()
            if isinstance(v, (list, tuple)):
                return [walk(u) for u in v]
            if isinstance(v, dict):
                return {k: walk(u) for k, u in v.items()}
            return v if isinstance(v, array) else {}
        return {k: walk(v) for k, v in self._public().items() if self._is_param_container(v)}

    def trainable_parameters(self):
        return self.parameters()

    def children(self):
        return {k: v for k, v in self._public().items() if self._is_param_container(v) and not isinstance(v, array)}

    def named_modules(self):
        out = [("", self)]

        def walk(prefix, v):
            if isinstance(v, Module):
                for n, m in v.named_modules():
                    out.append((f"{prefix}.{n}" if n else prefix, m))
            elif isinstance(v, (list, tuple)):
                for i, u in enumerate(v):
                    walk(f"{prefix}.{i}", u)
            elif isinstance(v, dict):
                for k, u in v.items():
                    walk(f"{prefix}.{k}", u)
        for k, v in self._public().items():
            walk(k, v)
        return out

    def modules(self):
        return [m for _, m in self.named_modules()]

    def _set_path(self, path, value):
        obj = self
        parts = path.split(".")
        for p in parts[:-1]:
            obj = obj[int(p)] if isinstance(obj, (list, tuple)) else obj[p] if isinstance(obj, dict) else getattr(obj, p)
        last = parts[-1]
        if isinstance(obj, list):
            obj[int(last)] = value
        elif isinstance(obj, dict):
            obj[last] = value
        else:
            if not hasattr(obj, last):
                raise ValueError(f"Module does not have parameter named \"{path}\".")
            setattr(obj, last, value)

    def load_weights(self, file_or_weights, strict=True):
        weights = list(load(file_or_weights).items()) if isinstance(file_or_weights, str) else list(file_or_weights)
        if strict:
            have = dict(tree_flatten(self.parameters()))
            new = dict(weights)
            if set(have) != set(new):
                raise ValueError(f"load_weights(strict): missing {sorted(set(have) - set(new))[:5]} extra {sorted(set(new) - set(have))[:5]}")
            for k, v in new.items():
                if tuple(v.shape) != tuple(have[k].shape):
                    raise ValueError(f"Expected shape {have[k].shape} but received shape {v.shape} for parameter {k}")
        for k, v in weights:
            self._set_path(k, v)
        return self

    def update(self, tree):
        for k, v in tree_flatten(tree):
            self._set_path(k, v)
        return self

    def update_modules(self, tree):
        for k, v in tree_flatten(tree, is_leaf=lambda m: isinstance(m, Module)):
            self._set_path(k, v)
        return self

    def apply(self, fn, filter_fn=None):
        self.update(tree_map(fn, self.parameters()))
        return self

    def set_dtype(self, dtype):
        return self.apply(lambda x: x.astype(dtype) if x._t.is_floating_point() else x)

    def eval(self):
        for m in self.modules():
            m._training = False
        return self

    def train(self, mode=True):
        for m in self.modules():
            m._training = mode
        return self

    def freeze(self, **kw):
        return self

    def unfreeze(self, **kw):
        return self

    @property
    def training(self):
        return self._training


builtins_any = _b.any


class Linear(Module):
    def __init__(self, input_dims, output_dims, bias=True):
        super().__init__()
        self.weight = zeros((output_dims, input_dims))
        if bias:
            self.bias = zeros((output_dims,))

    def __call__(self, x):
        if "bias" in vars(self):
            return addmm(self.bias, x, self.weight.T)
        return matmul(x, self.weight.T)

    def to_quantized(self, group_size=64, bits=4):
        return QuantizedLinear.from_linear(self, group_size, bits)


class Embedding(Module):
    def __init__(self, num_embeddings, dims):
        super().__init__()
        self.weight = zeros((num_embeddings, dims))

    def to_quantized(self, group_size=64, bits=4):
        return QuantizedEmbedding.from_embedding(self, group_size, bits)

    def __call__(self, x):
        return self.weight[x]

    def as_linear(self, x):
        return matmul(x, self.weight.T)


class RMSNorm(Module):
    def __init__(self, dims, eps=1e-5):
        super().__init__()
        self.weight = ones((dims,))
        self.eps = eps

    def __call__(self, x):
        return fast_rms_norm(x, self.weight, self.eps)


class LayerNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=True, bias=True):
        super().__init__()
        self.eps, self.dims = eps, dims
        if affine:
            self.weight = ones((dims,))
            if bias:
                self.bias = zeros((dims,))

    def __call__(self, x):
        return fast_layer_norm(x, vars(self).get("weight"), vars(self).get("bias"), self.eps)


class Conv2d(Module):
    """NHWC input, weight [O, kH, kW, I] (MLX layout)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.weight = zeros((out_channels, k[0], k[1], in_channels))
        if bias:
            self.bias = zeros((out_channels,))
        self.stride, self.padding, self.dilation = stride, padding, dilation

    def __call__(self, x):
        xt, w = x._t, self.weight._t
        out = torch.promote_types(xt.dtype, w.dtype)
        y = torch.nn.functional.conv2d(xt.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=self.stride,
                                       padding=self.padding, dilation=self.dilation).permute(0, 2, 3, 1)
        if "bias" in vars(self):
            y = y + self.bias._t.float()
        return array(y.to(out))


class GELU(Module):
    def __init__(self, approx="none"):
        super().__init__()
        self._approx = approx

    def __call__(self, x):
        if self._approx == "none":
            return gelu(x)
        return gelu_fast_approx(x) if self._approx == "fast" else gelu_approx(x)


class Dropout(Module):
    def __init__(self, p=0.5):
        super().__init__()
        self._p = p

    def __call__(self, x):
        if self._p == 0 or not self._training:
            return x
        raise NotImplementedError("mlx shim: dropout with p > 0 in training mode")


class QuantizedLinear(Module):
    def __init__(self, input_dims, output_dims, bias=True, group_size=64, bits=4):
        super().__init__()
        self.group_size, self.bits = group_size, bits
        self.weight, self.scales, self.biases = quantize(zeros((output_dims, input_dims)), group_size, bits)
        if bias:
            self.bias = zeros((output_dims,))

    def __call__(self, x):
        y = quantized_matmul(x, self.weight, self.scales, self.biases, True, self.group_size, self.bits)
        return y + self.bias if "bias" in vars(self) else y

    @classmethod
    def from_linear(cls, lin, group_size=64, bits=4):
        out_d, in_d = lin.weight.shape
        q = cls.__new__(cls)
        Module.__init__(q)
        q.group_size, q.bits = group_size, bits
        q.weight, q.scales, q.biases = quantize(lin.weight, group_size, bits)
        if "bias" in vars(lin):
            q.bias = lin.bias
        return q


class QuantizedEmbedding(Module):
    def __init__(self, num_embeddings, dims, group_size=64, bits=4):
        super().__init__()
        self.group_size, self.bits = group_size, bits
        self.weight, self.scales, self.biases = quantize(zeros((num_embeddings, dims)), group_size, bits)

    def __call__(self, x):
        idx = _index(x) if isinstance(x, array) else x
        return dequantize(array(self.weight._t[idx].reshape(-1, self.weight.shape[1])), array(self.scales._t[idx].reshape(-1, self.scales.shape[1])),
                          array(self.biases._t[idx].reshape(-1, self.biases.shape[1])), self.group_size, self.bits).reshape(*x.shape, -1)

    @classmethod
    def from_embedding(cls, emb, group_size=64, bits=4):
        q = cls.__new__(cls)
        Module.__init__(q)
        q.group_size, q.bits = group_size, bits
        q.weight, q.scales, q.biases = quantize(emb.weight, group_size, bits)
        return q


def nn_quantize(model, group_size=64, bits=4, class_predicate=None):
    """nn.quantize: every module with `to_quantized` (Linear, Embedding) for which the predicate holds is replaced in place."""
    class_predicate = class_predicate or (lambda _, m: hasattr(m, "to_quantized"))

    def walk(parent, key, v, path):
        if isinstance(v, Module):
            if class_predicate(path, v) and hasattr(v, "to_quantized"):
                new = v.to_quantized(group_size, bits)
                if isinstance(parent, (list, dict)):
                    parent[key] = new
                else:
                    setattr(parent, key, new)
                return
            for k2, u in list(v._public().items()):
                walk(v, k2, u, f"{path}.{k2}" if path else k2)
        elif isinstance(v, list):
            for i, u in enumerate(v):
                walk(v, i, u, f"{path}.{i}")
        elif isinstance(v, dict):
            for k2, u in list(v.items()):
                walk(v, k2, u, f"{path}.{k2}")
    for k, v in list(model._public().items()):
        walk(model, k, v, k)
    return model


def gelu(x):
    """exact erf GELU: x * (1 + erf(x / sqrt(2))) / 2, evaluated in fp32, one rounding."""
    t = x._t
    return array(torch.nn.functional.gelu(t.float()).to(t.dtype))


def gelu_approx(x):
    t = x._t
    return array(torch.nn.functional.gelu(t.float(), approximate="tanh").to(t.dtype))


def gelu_fast_approx(x):
    return x * sigmoid(1.702 * x)


def silu(x):
    return x * sigmoid(x)


def relu(x):
    return maximum(x, 0)


def log_softmax(x, axis=-1):
    return x - logsumexp(x, axis=axis, keepdims=True)


def nn_softmax(x, axis=-1):
    return softmax(x, axis)


def fast_rms_norm(x, weight, eps):
    t = x._t
    xf = t.float()
    n = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(t.dtype)
    return array(n) * weight


def fast_layer_norm(x, weight, bias, eps):
    t = x._t
    xf = t.float()
    mu = xf.mean(-1, keepdim=True)
    var = (xf - mu).pow(2).mean(-1, keepdim=True)
    y = array(((xf - mu) * torch.rsqrt(var + eps)).to(t.dtype))
    if weight is not None:
        y = y * weight
    if bias is not None:
        y = y + bias
    return y


def fast_sdpa(q, k, v, *, scale, mask=None):
    qt, kt, vt = q._t, k._t, v._t
    out = torch.promote_types(torch.promote_types(qt.dtype, kt.dtype), vt.dtype)
    w = (qt.float() * scale) @ kt.float().transpose(-1, -2)
    if mask is not None:
        w = w + mask._t.float()
    p = softmax(array(w), axis=-1)._t
    return array((p @ vt.float()).to(out))


def fast_rope(*a, **k):
    raise NotImplementedError("mlx shim: mx.fast.rope is off the pinned path")


# ---------------------------------------------------------------------------------------------------------------------
# installation
# ---------------------------------------------------------------------------------------------------------------------
def install():
    """Put `mlx`, `mlx.core`, `mlx.nn`, `mlx.utils`, `mlx.optimizers` (inert) into sys.modules.  Idempotent."""
    if "mlx" in sys.modules and getattr(sys.modules["mlx"], "__p3v_shim__", False):
        return sys.modules["mlx.core"], sys.modules["mlx.nn"]
    me = sys.modules[__name__]

    def mk(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m
    mlx, core, nn, utils, optim = (mk(n) for n in ("mlx", "mlx.core", "mlx.nn", "mlx.utils", "mlx.optimizers"))
    mlx.__p3v_shim__ = True
    core_names = ["array", "Dtype", "float32", "float16", "bfloat16", "int8", "int16", "int32", "int64", "uint8", "uint32", "bool_",
                  "inf", "nan", "pi", "eval", "compile", "zeros", "ones", "full", "zeros_like", "ones_like", "arange", "linspace",
                  "matmul", "addmm", "concatenate", "stack", "split", "repeat", "tile", "broadcast_to", "expand_dims", "squeeze",
                  "flatten", "reshape", "transpose", "swapaxes", "pad", "triu", "tril", "where", "sum", "mean", "max", "min", "all",
                  "any", "argmax", "argmin", "argsort", "argpartition", "sort", "exp", "log", "cos", "sin", "sqrt", "rsqrt", "tanh",
                  "erf", "sigmoid", "abs", "square", "maximum", "minimum", "isinf", "isnan", "stop_gradient", "logsumexp", "softmax",
                  "take", "load", "save_safetensors", "quantize", "dequantize", "quantized_matmul", "round", "clip"]
    for n in core_names:
        setattr(core, n, getattr(me, n))
    fast = mk("mlx.core.fast")
    fast.rms_norm, fast.layer_norm, fast.scaled_dot_product_attention, fast.rope = fast_rms_norm, fast_layer_norm, fast_sdpa, fast_rope
    core.fast = fast
    rnd = mk("mlx.core.random")
    _gen = torch.Generator().manual_seed(0)
    rnd.seed = lambda s: _gen.manual_seed(int(s))
    rnd.uniform = lambda low=0.0, high=1.0, shape=(), dtype=float32: array((torch.rand(tuple(shape), generator=_gen) * (high - low) + low).to(dtype.t))
    rnd.normal = lambda shape=(), dtype=float32, loc=0.0, scale=1.0: array((torch.randn(tuple(shape), generator=_gen) * scale + loc).to(dtype.t))
    core.random = rnd
    for n in ("Module", "Linear", "Embedding", "RMSNorm", "LayerNorm", "Conv2d", "GELU", "Dropout", "QuantizedLinear", "QuantizedEmbedding", "gelu", "gelu_approx",
              "gelu_fast_approx", "silu", "relu", "log_softmax"):
        setattr(nn, n, getattr(me, n))
    nn.softmax = nn_softmax
    nn.quantize = nn_quantize
    nn.value_and_grad = _unsupported("nn.value_and_grad")
    nn.losses = types.SimpleNamespace(cross_entropy=_unsupported("nn.losses.cross_entropy"))
    nn.functional = types.SimpleNamespace()
    utils.tree_flatten, utils.tree_unflatten, utils.tree_map = tree_flatten, tree_unflatten, tree_map
    mlx.core, mlx.nn, mlx.utils, mlx.optimizers = core, nn, utils, optim

    def _optim_attr(k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _unsupported(f"mlx.optimizers.{k}")
    optim.__getattr__ = _optim_attr
    return core, nn
