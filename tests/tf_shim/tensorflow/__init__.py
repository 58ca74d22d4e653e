"""
TEST-ONLY stand-in for the handful of TensorFlow names that tfrt/geometry.py uses, on torch-CPU
float64 tensors, so that the reference's OWN source file can be executed in the build container
(where TensorFlow is not installed) by tests/golden/make_reference_golden.py.  Never imported by
the product package, the GPU tests or bench.py; it does not travel anywhere that matters (the
reference does not exist on the GPU box).

Every op is one correctly rounded float64 operation, like TensorFlow's CPU kernels (Eigen): in
particular ``sqrt`` is numpy's IEEE square root, not torch's (see oracle/geom.py).  Only eager
semantics are provided: ``tf.function`` returns the undecorated function.
"""
import builtins as _builtins
import contextlib

import numpy as np
import torch

float64 = torch.float64
float32 = torch.float32
int64 = torch.int64
int32 = torch.int32
bool = torch.bool  # noqa: A001


class TensorSpec:
    def __init__(self, shape=None, dtype=None, name=None):
        self.shape, self.dtype, self.name = shape, dtype, name


def function(func=None, input_signature=None, **_):
    if func is None:
        return lambda f: f
    return func


@contextlib.contextmanager
def name_scope(_name):
    yield


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, (_builtins.bool, np.bool_)):
        return torch.tensor(_builtins.bool(x))
    if isinstance(x, np.ndarray) and x.dtype.kind in "biu":
        return torch.as_tensor(x)              # masks and indices keep their type
    return torch.as_tensor(x, dtype=torch.float64)


def constant(x, dtype=None, name=None):
    return torch.as_tensor(x, dtype=dtype or torch.float64)


def cast(x, dtype, name=None):
    return _t(x).to(dtype)


def meshgrid(*args, indexing="xy", name=None):
    return list(torch.meshgrid(*[_t(a) for a in args], indexing=indexing))


def where(cond, x=None, y=None, name=None):
    x, y = _t(x), _t(y)
    return torch.where(cond, x, y)


def ones_like(x, dtype=None, name=None):
    return torch.ones_like(_t(x), dtype=dtype)


def zeros_like(x, dtype=None, name=None):
    return torch.zeros_like(_t(x), dtype=dtype)


def broadcast_to(x, shape, name=None):
    return torch.broadcast_to(_t(x), tuple(int(s) for s in shape))


def reshape(x, shape, name=None):
    return torch.reshape(_t(x), tuple(shape))


def shape(x, out_type=torch.int32, name=None):
    return torch.tensor(tuple(_t(x).shape), dtype=torch.int64)


def stack(xs, axis=0, name=None):
    return torch.stack([_t(x) for x in xs], dim=axis)


def unstack(x, num=None, axis=0, name=None):
    return list(torch.unbind(_t(x), dim=axis))


def reduce_sum(x, axis=None, keepdims=False, name=None):
    return torch.sum(_t(x)) if axis is None else torch.sum(_t(x), dim=axis, keepdim=keepdims)


def abs(x, name=None):  # noqa: A001
    return torch.abs(_t(x))


def sign(x, name=None):
    return torch.sign(_t(x))


class _Sqrt(torch.autograd.Function):
    """Correctly rounded float64 square root (numpy) with the usual derivative."""

    @staticmethod
    def forward(ctx, x):
        y = torch.from_numpy(np.sqrt(x.detach().contiguous().numpy()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g / (2 * y)


def sqrt(x, name=None):
    return _Sqrt.apply(_t(x))


def sin(x, name=None):
    return torch.sin(_t(x))


def cos(x, name=None):
    return torch.cos(_t(x))


def asin(x, name=None):
    return torch.asin(_t(x))


def atan2(y, x, name=None):
    return torch.atan2(_t(y), _t(x))


def less(a, b, name=None):
    return _t(a) < _t(b)


def less_equal(a, b, name=None):
    return _t(a) <= _t(b)


def greater(a, b, name=None):
    return _t(a) > _t(b)


def greater_equal(a, b, name=None):
    return _t(a) >= _t(b)


def equal(a, b, name=None):
    return _t(a) == _t(b)


def not_equal(a, b, name=None):
    return _t(a) != _t(b)


def logical_and(a, b, name=None):
    return torch.logical_and(a, b)


def logical_or(a, b, name=None):
    return torch.logical_or(a, b)


def logical_not(a, name=None):
    return torch.logical_not(a)


class _Math:
    @staticmethod
    def mod(x, y, name=None):
        return torch.remainder(_t(x), _t(y))          # tf.math.mod is floor-mod

    floormod = mod

    @staticmethod
    def l2_normalize(x, axis=None, epsilon=1e-12, name=None):
        x = _t(x)
        sq = torch.sum(x * x, dim=axis, keepdim=True)
        return x * torch.rsqrt(torch.clamp(sq, min=epsilon))   # torch.rsqrt is IEEE 1/sqrt here

    is_finite = staticmethod(lambda x: torch.isfinite(_t(x)))
    squared_difference = staticmethod(lambda a, b: (_t(a) - _t(b)) ** 2)


math = _Math()


# ------------------------------------------------------------------- used by tfrt/engine.py,
# tfrt/operation.py, tfrt/materials.py (tests/golden/make_reference_trace_golden.py)

DType = torch.dtype
dtype = torch.dtype


class TensorShape(tuple):
    pass


def concat(values, axis=0, name=None):
    return torch.cat([_t(v) for v in values], dim=axis)


def zeros(shape, dtype=torch.float32, name=None):
    return torch.zeros(tuple(shape), dtype=dtype)


def ones(shape, dtype=torch.float32, name=None):
    return torch.ones(tuple(shape), dtype=dtype)


def fill(dims, value, name=None):
    return torch.full(tuple(int(d) for d in dims), float(value), dtype=torch.float64)


def range(start, limit=None, delta=1, dtype=None, name=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    return torch.arange(int(start), int(limit), int(delta), dtype=dtype or torch.int64)


def transpose(a, perm=None, name=None):
    a = _t(a)
    return a.t() if perm is None else a.permute(*perm)


def gather(params, indices, axis=0, name=None):
    assert axis == 0
    return _t(params)[_t(indices).to(torch.int64)]


def gather_nd(params, indices, name=None):
    idx = _t(indices).to(torch.int64)
    return _t(params)[tuple(idx[..., k] for k in builtins_range(idx.shape[-1]))]


def boolean_mask(tensor, mask, axis=None, name=None):
    return _t(tensor)[mask]                    # keeps the order (stable), like tf.boolean_mask


def argmin(input, axis=None, output_type=torch.int64, name=None):  # noqa: A002
    # tf.argmin returns the FIRST index of the minimum; torch.argmin does not promise that on
    # ties, so spell it out: min value, then the first position that attains it
    x = _t(input)
    m = torch.min(x, dim=axis, keepdim=True).values
    n = x.shape[axis]
    pos = torch.arange(n, dtype=torch.int64).reshape([-1 if k == axis else 1 for k in builtins_range(x.dim())])
    first = torch.where(x == m, pos, torch.full_like(pos, n)).min(dim=axis).values
    return first.to(output_type)


def reduce_max(x, axis=None, keepdims=False, name=None):
    return torch.max(_t(x)) if axis is None else torch.max(_t(x), dim=axis, keepdim=keepdims).values


def reduce_min(x, axis=None, keepdims=False, name=None):
    return torch.min(_t(x)) if axis is None else torch.min(_t(x), dim=axis, keepdim=keepdims).values


def reduce_any(x, axis=None, keepdims=False, name=None):
    return torch.any(x) if axis is None else torch.any(x, dim=axis, keepdim=keepdims)


def reduce_prod(x, axis=None, name=None):
    return torch.prod(_t(x)) if axis is None else torch.prod(_t(x), dim=axis)


def expand_dims(x, axis, name=None):
    return torch.unsqueeze(_t(x), axis)


def squeeze(x, axis=None, name=None):
    return torch.squeeze(_t(x)) if axis is None else torch.squeeze(_t(x), axis)


import builtins as _builtins  # noqa: E402

builtins_range = _builtins.range


# ------------------------------------------------------------------- used by tfrt/sources.py,
# tfrt/distributions.py, tfrt/boundaries.py (tests/golden/make_reference_host_golden.py)

string = str


def rank(x, name=None):
    return _t(x).dim()


def convert_to_tensor(x, dtype=None, name=None):
    t = _t(x)
    return t if dtype is None else t.to(dtype)


def linspace(start, stop, num, name=None, axis=0):
    return torch.linspace(float(start), float(stop), int(num), dtype=torch.float64)


def acos(x, name=None):
    return torch.acos(_t(x))


def norm(x, ord="euclidean", axis=None, keepdims=False, name=None):  # noqa: A002
    x = _t(x)
    sq = torch.sum(x * x) if axis is None else torch.sum(x * x, dim=axis, keepdim=keepdims)
    return sqrt(sq)


def reduce_mean(x, axis=None, keepdims=False, name=None):
    return torch.mean(_t(x)) if axis is None else torch.mean(_t(x), dim=axis, keepdim=keepdims)


def pad(tensor, paddings, mode="CONSTANT", constant_values=0, name=None):
    t = _t(tensor)
    p = torch.as_tensor(paddings).to(torch.int64).reshape(-1, 2).tolist()
    flat = []
    for before, after in reversed(p):
        flat += [int(before), int(after)]
    return torch.nn.functional.pad(t, flat, value=constant_values)


def repeat(input, repeats, axis=None, name=None):  # noqa: A002
    return torch.repeat_interleave(_t(input), torch.as_tensor(repeats), dim=axis)


def stop_gradient(x, name=None):
    return _t(x).detach()


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return torch.clamp(_t(t), min=float(clip_value_min), max=float(clip_value_max))


def Variable(initial_value, dtype=None, trainable=True, validate_shape=True, name=None):
    t = _t(initial_value).detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    t.requires_grad_(_builtins.bool(trainable) and t.is_floating_point())

    def assign(value, t=t):
        with torch.no_grad():
            t.copy_(_t(value))
        return t

    def assign_add(value, t=t):
        with torch.no_grad():
            t.add_(_t(value))
        return t

    def assign_sub(value, t=t):
        with torch.no_grad():
            t.sub_(_t(value))
        return t

    t.assign, t.assign_add, t.assign_sub = assign, assign_add, assign_sub
    return t


class _Linalg:
    @staticmethod
    def normalize(tensor, ord="euclidean", axis=None, name=None):  # noqa: A002
        x = _t(tensor)
        n = sqrt(torch.sum(x * x, dim=axis, keepdim=True))
        return x / n, n

    @staticmethod
    def cross(a, b, name=None):
        a, b = _t(a), _t(b)
        return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                            a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                            a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


linalg = _Linalg()


class _Errors:
    class InvalidArgumentError(Exception):
        pass


errors = _Errors()


class _Debugging:
    @staticmethod
    def _ok(*a, **k):
        return None

    assert_integer = assert_greater = assert_greater_equal = assert_less = _ok
    assert_less_equal = assert_positive = assert_equal = _ok


debugging = _Debugging()


class _Random:
    @staticmethod
    def uniform(shape, minval=0, maxval=None, dtype=torch.float64, seed=None, name=None):
        maxval = 1.0 if maxval is None else maxval
        return torch.rand(tuple(int(s) for s in shape), dtype=dtype) * (maxval - minval) + minval

    @staticmethod
    def normal(shape, mean=0.0, stddev=1.0, dtype=torch.float64, seed=None, name=None):
        return torch.randn(tuple(int(s) for s in shape), dtype=dtype) * stddev + mean


random = _Random()
_Math.floormod = staticmethod(lambda x, y, name=None: torch.remainder(_t(x), _t(y)))


# tf tensors hand out .numpy() whether or not a tape watches them; torch refuses for tensors that
# require grad.  The reference calls .numpy() on such tensors (boundaries.py:932), so inside the
# generator processes -- the only ones that import this stand-in -- numpy() detaches first.
_torch_numpy = torch.Tensor.numpy


def _numpy_like_tf(self, *args, **kwargs):
    return _torch_numpy(self.detach().cpu(), *args, **kwargs)


torch.Tensor.numpy = _numpy_like_tf


# ------------------------------------------------------------------- used by tfrt/optimizer.py
# (tests/golden/make_reference_optimizer_golden.py)

def matmul(a, b, name=None):
    return torch.matmul(_t(a), _t(b))


class GradientTape:
    """tape.gradient(target, sources) = torch.autograd.grad of sum(target): TensorFlow sums a
    non-scalar target the same way; unconnected sources come back as None."""

    def __init__(self, persistent=False, watch_accessed_variables=True):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, _tensor):
        pass

    def gradient(self, target, sources):
        single = isinstance(sources, torch.Tensor)
        srcs = [sources] if single else list(sources)
        t = _t(target)
        if not t.requires_grad:
            out = [None] * len(srcs)
        else:
            out = list(torch.autograd.grad(t.sum(), srcs, allow_unused=True, retain_graph=True))
        return out[0] if single else out


class _SGD:
    """Keras OptimizerV2 SGD as tfrt/optimizer.py uses it: built with the defaults (learning rate
    0.01, momentum 0.0) and ``nesterov=True``; the momentum branch is chosen at CONSTRUCTION
    (``self._momentum = momentum > 0``), so assigning ``opt.momentum`` afterwards, which is all the
    reference does (optimizer.py:128-132), leaves the update at ``var -= learning_rate * grad``."""

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, name="SGD", **kwargs):
        self.learning_rate = learning_rate
        self.momentum = momentum
        self.nesterov = nesterov
        self._use_momentum = momentum > 0

    def apply_gradients(self, grads_and_vars, name=None):
        assert not self._use_momentum
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is not None:
                    v.sub_(self.learning_rate * _t(g))


class _Optimizers:
    SGD = _SGD


optimizers = _Optimizers()
