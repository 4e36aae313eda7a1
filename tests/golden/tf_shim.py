"""A small lazy-graph stand-in for the slice of TensorFlow 1.x that cocosci/NSC uses.  TEST INFRASTRUCTURE ONLY.

Purpose: TensorFlow is absent from the build container, so the reference's *own Python code*
(``nn_core_operator.py``, ``loss_terms_and_measures.py``, the builders / trainers of
``neural_speech_coding_module.py`` and ``cmrl.py``) cannot run as shipped.  This module gives it a ``tf``
object that builds a lazy expression graph (placeholders, variables, variable scopes, ``Session.run`` with a
feed dict, ``AdamOptimizer.minimize``, ``Saver``) and evaluates it in float64 on PyTorch-CPU, with autograd
supplying ``tf.gradients``.  ``tests/golden/make_reference_exec.py`` imports the reference from
``/root/reference`` on top of it, RUNS the reference's functions and commits their outputs as
``tests/golden/reference_exec.npz``.  Nothing here or in the reference travels to the GPU box; only the
fixture does.

What is the reference's own code when run this way: every composition - the quantizer formula, the gated block,
the encoder / decoder / cascade wiring, the four losses and their assembly into the two optimised [B]-vector
losses per phase, which variables each optimizer owns, the training loops, the tau controllers, the validation
loop, journal lines and checkpoint names.

What is restated here (the primitives; TensorFlow's source is not under /root/reference - [TF-semantics]):
  * ``layers.conv1d`` / ``keras.layers.SeparableConv1D``: cross-correlation, kernel [K,Cin,Cout], SAME padding with
    the odd sample on the right, glorot-uniform kernels and zero biases, variable names ``<scope>/conv1d[_n]/kernel``;
  * ``nn.leaky_relu`` (alpha 0.2), ``nn.softmax`` (last axis), ``nn.top_k`` (k=1, lowest index among ties), ``one_hot``;
  * ``signal.stft(window_fn=None)`` = framed rFFT, ``signal.linear_to_mel_weight_matrix`` (HTK mel, float64 inside, returned
    as float32);
  * ``train.AdamOptimizer`` (TF1 form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps added to sqrt(v)); ``minimize`` of a
    non-scalar loss differentiates its SUM (``tf.gradients`` seeds ones);
  * elementwise / reduce / reshape ops, broadcasting, ``cond``, ``py_func`` (inputs handed over as float32 arrays, as
    TF would).
Arithmetic is float64 (TF computes these graphs in float32): the fixture is the exact value the float32 graph
approximates, which is what a 1e-4 parity bound needs.

Variable initial values are a deterministic function of the variable's NAME (``name_seeded_uniform``) so that any
implementation can regenerate the same float32-representable weights without shipping 350 k numbers per codec.
"""
from __future__ import annotations

import contextlib
import math
import os
import re
import types
import zlib

import numpy as np
import torch

DT = torch.float64
TRACE_B = 3  # stand-in for the unknown (None) batch size while shapes are traced; no layer has a dimension of 3


# ------------------------------------------------------------------------------------------------
# deterministic name-seeded initial values (shared recipe: tests/_util.py::name_seeded_uniform)
# ------------------------------------------------------------------------------------------------
def name_seeded_uniform(name, shape, lim, seed=20200504):
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    return rng.uniform(-lim, lim, size=shape).astype(np.float32).astype(np.float64)


# ------------------------------------------------------------------------------------------------
# graph state
# ------------------------------------------------------------------------------------------------
class Graph:
    def __init__(self):
        self.variables = []          # all Variable nodes in creation order (global_variables)
        self.scope_stack = []
        self.layer_counts = {}
        self.var_names = {}

    def as_default(self):
        return _GraphCtx(self)

    def scope_prefix(self):
        return "/".join(self.scope_stack) + ("/" if self.scope_stack else "")

    def unique_layer(self, base):
        key = (self.scope_prefix(), base)
        n = self.layer_counts.get(key, 0)
        self.layer_counts[key] = n + 1
        return self.scope_prefix() + (base if n == 0 else f"{base}_{n}")

    def unique_var(self, full):
        n = self.var_names.get(full, 0)
        self.var_names[full] = n + 1
        return full if n == 0 else f"{full}_{n}"


class _GraphCtx:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        _STATE["stack"].append(self.g)
        return self.g

    def __exit__(self, *a):
        _STATE["stack"].pop()


_STATE = {"stack": [Graph()]}


def _g() -> Graph:
    return _STATE["stack"][-1]


def reset_default_graph():
    _STATE["stack"][-1] = Graph()


# ------------------------------------------------------------------------------------------------
# nodes
# ------------------------------------------------------------------------------------------------
class Shape(tuple):
    def as_list(self):
        return list(self)


def _to_t(v):
    if isinstance(v, torch.Tensor):
        return v
    a = np.asarray(v)
    if a.dtype == np.bool_:
        return torch.tensor(a)
    if np.iscomplexobj(a):
        return torch.tensor(a, dtype=torch.complex128)
    if a.dtype.kind in "iu":
        return torch.tensor(a, dtype=torch.int64)
    return torch.tensor(a.astype(np.float64), dtype=DT)


class Node:
    """A lazy tensor.  ``fn(*input_values) -> torch.Tensor``."""
    __array_ufunc__ = None       # numpy scalars defer to our reflected operators
    __array_priority__ = 1000

    def __init__(self, fn, inputs=(), name=None, batch_dep=None, trace=None):
        self.fn = fn
        self.inputs = tuple(inputs)
        self.name = name
        self.batch_dep = any(i.batch_dep for i in self.inputs) if batch_dep is None else batch_dep
        with torch.no_grad():
            self.trace = fn(*[i.trace for i in self.inputs]) if trace is None else trace

    # ---- static shape, as TF would report it (None for the batch axis) ----
    @property
    def shape(self):
        s = list(self.trace.shape)
        if self.batch_dep and s and s[0] == TRACE_B:
            s[0] = None
        return Shape(s)

    def get_shape(self):
        return self.shape

    def eval(self, feed_dict=None, session=None):
        return Session().run(self, feed_dict)

    # ---- operators ----
    def _bin(self, other, f, rev=False):
        o = convert(other)
        a, b = (o, self) if rev else (self, o)
        return Node(f, (a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.div)
    def __rtruediv__(self, o): return self._bin(o, torch.div, True)
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __neg__(self): return Node(torch.neg, (self,))

    def __getitem__(self, idx):
        return Node(lambda v: v[idx], (self,))

    def __iter__(self):
        raise TypeError("a lazy tensor is not iterable")

    def __bool__(self):
        raise TypeError("a lazy tensor has no truth value at graph-construction time")

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other


class Placeholder(Node):
    def __init__(self, dtype, shape, name):
        batch = shape is not None and len(shape) > 0 and shape[0] is None
        if shape is None:
            shp = ()
        else:
            shp = tuple(TRACE_B if d is None else int(d) for d in shape)
        tr = torch.zeros(shp, dtype=torch.bool) if dtype == "bool" else torch.zeros(shp, dtype=DT)
        self.dtype = dtype
        self.declared_shape = shape
        super().__init__(None, (), name=name, batch_dep=batch, trace=tr)


class Variable(Node):
    def __init__(self, initial_value, dtype=None, name=None, trainable=True, _full_name=None, _init_fn=None):
        g = _g()
        full = _full_name if _full_name is not None else g.unique_var(g.scope_prefix() + (name or "Variable"))
        self.var_name = full
        self.trainable = bool(trainable)
        self._init_fn = _init_fn if _init_fn is not None else (
            lambda: np.array(np.asarray(initial_value, dtype=np.float32), dtype=np.float64))
        self.value = None            # torch leaf, set by an initializer / Saver.restore
        super().__init__(None, (), name=full + ":0", batch_dep=False, trace=_to_t(self._init_fn()))
        g.variables.append(self)

    def initialize(self):
        self.value = _to_t(self._init_fn()).clone().requires_grad_(True)

    def assign_numpy(self, a):
        self.value = _to_t(np.asarray(a, np.float64)).clone().requires_grad_(True)

    def numpy(self):
        return self.value.detach().numpy().copy()


def convert(v):
    if isinstance(v, Node):
        return v
    if isinstance(v, (list, tuple)) and any(isinstance(e, Node) for e in v):
        elems = [convert(e) for e in v]
        return Node(lambda *a: torch.stack([x.to(DT) if not x.is_complex() else x for x in a], 0), elems)
    if isinstance(v, np.ndarray) and v.dtype == object:
        return convert(list(v))
    t = _to_t(v)
    return Node(None, (), batch_dep=False, trace=t)


def _const_value(n):
    return n.trace


# ------------------------------------------------------------------------------------------------
# evaluation
# ------------------------------------------------------------------------------------------------
def evaluate(fetch_nodes, feeds):
    """Iterative post-order evaluation with memoisation.  ``feeds``: {Node: torch value} (any node may be fed)."""
    memo = dict(feeds)
    for root in fetch_nodes:
        stack = [(root, False)]
        while stack:
            n, done = stack.pop()
            if n in memo:
                continue
            if isinstance(n, Placeholder):
                raise ValueError(f"placeholder {n.name} was not fed")
            if isinstance(n, Variable):
                if n.value is None:
                    raise ValueError(f"variable {n.var_name} is uninitialised")
                memo[n] = n.value
                continue
            if n.fn is None:           # constant
                memo[n] = n.trace
                continue
            if done:
                memo[n] = n.fn(*[memo[i] for i in n.inputs])
            else:
                stack.append((n, True))
                for i in n.inputs:
                    if i not in memo:
                        stack.append((i, False))
    return memo


class _Op:
    """A graph op with side effects (initializers, train steps)."""

    def run(self, session, feeds):
        raise NotImplementedError


class _InitOp(_Op):
    def __init__(self, variables):
        self.variables = list(variables)

    def run(self, session, feeds):
        for v in self.variables:
            v.initialize()


class Session:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None):
        feeds = {}
        for k, v in (feed_dict or {}).items():
            if isinstance(k, Placeholder) and k.dtype == "bool":
                feeds[k] = torch.tensor(bool(v))
            else:
                feeds[k] = _to_t(np.asarray(v, dtype=np.float64))
        flat = []

        def collect(f):
            if isinstance(f, (list, tuple)):
                for e in f:
                    collect(e)
            elif isinstance(f, Node):
                flat.append(f)

        collect(fetches)
        ops = []

        def collect_ops(f):
            if isinstance(f, (list, tuple)):
                for e in f:
                    collect_ops(e)
            elif isinstance(f, _Op):
                ops.append(f)

        collect_ops(fetches)
        with torch.no_grad():
            memo = evaluate(flat, feeds) if flat else {}
        for op in ops:
            op.run(self, feeds)

        def build(f):
            if isinstance(f, (list, tuple)):
                return [build(e) for e in f]
            if isinstance(f, Node):
                v = memo[f].detach()
                a = v.numpy()
                if a.dtype == np.float64:
                    a = a.astype(np.float64)
                return a.copy() if a.ndim else a[()]
            if isinstance(f, _Op):
                return None
            return f

        return build(fetches)


# ------------------------------------------------------------------------------------------------
# primitives ([TF-semantics] restatements)
# ------------------------------------------------------------------------------------------------
def _same_pad(T, k, d, s):
    t_out = -(-T // s)
    pad = max((t_out - 1) * s + (k - 1) * d + 1 - T, 0)
    return t_out, pad // 2, pad - pad // 2


def _conv1d_val(x, W, b, dil, stride):
    # x [B,T,Cin] channels_last, W [K,Cin,Cout]; cross-correlation, zero padding
    K = W.shape[0]
    t_out, pl, pr = _same_pad(x.shape[1], K, dil, stride)
    xb = torch.nn.functional.pad(x.transpose(1, 2), (pl, pr))
    y = torch.nn.functional.conv1d(xb, W.permute(2, 1, 0), b, stride=stride, dilation=dil)
    return y.transpose(1, 2)


def _act_node(node, activation):
    return node if activation is None else activation(node)


def _glorot_var(name, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    shp = tuple(int(s) for s in shape)
    return Variable(None, name=None, _full_name=name, _init_fn=lambda: name_seeded_uniform(name, shp, lim))


def _zeros_var(name, n):
    return Variable(None, name=None, _full_name=name, _init_fn=lambda: np.zeros(int(n)))


def layers_conv1d(inputs, filters, kernel_size, padding="valid", activation=None, dilation_rate=1, strides=1,
                  data_format="channels_last", **kw):
    assert padding.upper() == "SAME" and data_format == "channels_last"
    cin = int(inputs.shape[-1])
    K, cout = int(kernel_size), int(filters)
    lname = _g().unique_layer("conv1d")
    W = _glorot_var(lname + "/kernel", (K, cin, cout), K * cin, K * cout)
    b = _zeros_var(lname + "/bias", cout)
    d, s = int(dilation_rate), int(strides)
    y = Node(lambda x, w, bb: _conv1d_val(x, w, bb, d, s), (inputs, W, b))
    return _act_node(y, activation)


class SeparableConv1D:
    def __init__(self, filters, kernel_size, padding="valid", activation=None, dilation_rate=1, strides=1,
                 data_format="channels_last", **kw):
        assert padding.upper() == "SAME" and int(dilation_rate) == 1 and int(strides) == 1
        self.filters, self.k, self.activation = int(filters), int(kernel_size), activation

    def __call__(self, inputs):
        C = int(inputs.shape[-1])
        K, cout = self.k, self.filters
        lname = _g().unique_layer("separable_conv1d")
        Wd = _glorot_var(lname + "/depthwise_kernel", (K, C, 1), K * C, K)      # depth multiplier 1
        Wp = _glorot_var(lname + "/pointwise_kernel", (1, C, cout), C, cout)
        b = _zeros_var(lname + "/bias", cout)

        def f(x, wd, wp, bb):
            t_out, pl, pr = _same_pad(x.shape[1], K, 1, 1)
            xb = torch.nn.functional.pad(x.transpose(1, 2), (pl, pr))
            dw = torch.nn.functional.conv1d(xb, wd.permute(1, 2, 0), None, groups=C)
            return torch.nn.functional.conv1d(dw, wp.permute(2, 1, 0), bb).transpose(1, 2)

        return _act_node(Node(f, (inputs, Wd, Wp, b)), self.activation)


def _hertz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)


def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                lower_edge_hertz=125.0, upper_edge_hertz=3800.0, dtype=None):
    nyq = float(sample_rate) / 2.0
    lin = np.linspace(0.0, nyq, int(num_spectrogram_bins))[1:]
    spec_mel = _hertz_to_mel(lin)[:, None]
    edges = np.linspace(_hertz_to_mel(lower_edge_hertz), _hertz_to_mel(upper_edge_hertz), int(num_mel_bins) + 2)
    lower, center, upper = edges[:-2][None], edges[1:-1][None], edges[2:][None]
    w = np.maximum(0.0, np.minimum((spec_mel - lower) / (center - lower), (upper - spec_mel) / (upper - center)))
    w = np.pad(w, [[1, 0], [0, 0]]).astype(np.float32).astype(np.float64)
    return convert(w)


def stft(signals, frame_length, frame_step, fft_length=None, window_fn="hann", pad_end=False):
    assert window_fn is None, "the reference only calls stft(window_fn=None)"
    fl, fs, nfft = int(frame_length), int(frame_step), int(fft_length)

    def f(x):
        n = 1 + (x.shape[-1] - fl) // fs
        frames = torch.stack([x[..., i * fs:i * fs + fl] for i in range(n)], dim=-2)
        return torch.fft.rfft(frames, n=nfft, dim=-1)

    return Node(f, (convert(signals),))


def top_k(x, k=1):
    def idx(v):
        # "if two elements are equal, the lower-index element appears first" (tf.nn.top_k); np.argmax documents
        # first-occurrence semantics, torch.argmax does not
        return torch.from_numpy(np.argmax(v.detach().numpy(), axis=-1)).unsqueeze(-1)

    n = convert(x)
    return types.SimpleNamespace(indices=Node(idx, (n,)), values=Node(lambda v: v.max(-1, keepdim=True)[0], (n,)))


def one_hot(indices, depth):
    d = int(depth)
    return Node(lambda i: torch.nn.functional.one_hot(i.to(torch.int64), d).to(DT), (convert(indices),))


def reshape(x, shape):
    shp = tuple(int(s) if s is not None else -1 for s in shape)
    return Node(lambda v: v.reshape(shp), (convert(x),))


def _reduce(f):
    def op(input_tensor=None, axis=None, keepdims=False, **kw):
        n = convert(input_tensor)
        if axis is None:
            return Node(lambda v: f(v), (n,))
        return Node(lambda v: f(v, dim=axis, keepdim=keepdims), (n,))
    return op


def _un(f):
    return lambda x, name=None: Node(f, (convert(x),))


def _bi(f):
    return lambda a, b, name=None: Node(f, (convert(a), convert(b)))


def cast(x, dtype):
    return convert(x)


def cond(pred, true_fn, false_fn):
    t, f = convert(true_fn()), convert(false_fn())
    return Node(lambda p, a, b: a if bool(p) else b, (convert(pred), t, f))


def py_func(func, inp, Tout, **kw):
    nodes = [convert(i) for i in inp]
    n_out = len(Tout) if isinstance(Tout, (list, tuple)) else 1

    def call(*vals):
        args = []
        for n, v in zip(nodes, vals):
            a = v.detach().numpy()
            if a.dtype == np.float64 and (n.inputs or isinstance(n, (Placeholder, Variable))):
                a = a.astype(np.float32)      # TF hands float32 tensors to the Python function
            elif a.ndim == 0:
                a = a[()]
                if float(a) == int(a):
                    a = int(a)               # python ints passed as constants (e.g. the LPC order)
            args.append(a)
        out = func(*args)
        outs = out if isinstance(out, (list, tuple)) and n_out > 1 else [out]
        return [_to_t(np.asarray(o)) for o in outs]

    results = []
    cache = {}

    def make(i):
        def f(*vals):
            key = tuple(id(v) for v in vals)
            if cache.get("key") != key:
                cache["key"] = key
                cache["val"] = call(*vals)
            return cache["val"][i]
        return f

    for i in range(n_out):
        results.append(Node(make(i), nodes))
    return results


def custom_gradient(f):
    def wrapped(*a, **k):
        return f(*a, **k)[0]
    wrapped.__name__ = getattr(f, "__name__", "custom_gradient")
    return wrapped


# ------------------------------------------------------------------------------------------------
# scopes, collections, savers, optimizers
# ------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def variable_scope(name, *a, **k):
    g = _g()
    g.scope_stack.append(name)
    try:
        yield name
    finally:
        g.scope_stack.pop()


class GraphKeys:
    TRAINABLE_VARIABLES = "trainable_variables"
    GLOBAL_VARIABLES = "variables"


def trainable_variables(scope=None):
    return get_collection(GraphKeys.TRAINABLE_VARIABLES, scope)


def global_variables(scope=None):
    return get_collection(GraphKeys.GLOBAL_VARIABLES, scope)


def get_collection(key, scope=None):
    vs = _g().variables
    if key == GraphKeys.TRAINABLE_VARIABLES:
        vs = [v for v in vs if v.trainable]
    if scope is not None:
        vs = [v for v in vs if re.match(scope, v.name)]   # tf filters with re.match on the item name
    return list(vs)


def global_variables_initializer():
    return _InitOp(_g().variables)


def variables_initializer(var_list):
    return _InitOp(var_list)


CKPT_DIR_ENV = "NSC_SHIM_CKPT_DIR"
SAVE_LOG = []   # (path, [variable names]) for every Saver.save, in call order


class Saver:
    """Checkpoints as ``<path>.npz`` keyed by variable name (the reference's path strings are kept verbatim)."""

    def __init__(self, var_list=None, **k):
        self.var_list = list(var_list) if var_list is not None else list(_g().variables)

    @staticmethod
    def _file(path):
        root = os.environ.get(CKPT_DIR_ENV)
        p = path if root is None else os.path.join(root, path.lstrip("./"))
        os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
        return p + ".npz"

    def save(self, sess, path, **k):
        np.savez(self._file(path), **{v.var_name: v.numpy() for v in self.var_list})
        SAVE_LOG.append((path, [v.var_name for v in self.var_list]))
        return path

    def restore(self, sess, path):
        data = np.load(self._file(path))
        for v in self.var_list:
            if v.var_name not in data.files:
                raise KeyError(f"{v.var_name} not found in checkpoint {path}")
            v.assign_numpy(data[v.var_name])


TRAIN_HOOKS = []   # callables(train_op, feeds, loss_value, grads_by_name) run after every optimizer step


class _TrainOp(_Op):
    def __init__(self, opt, loss, var_list):
        self.opt, self.loss, self.var_list = opt, loss, list(var_list)

    def run(self, session, feeds):
        opt = self.opt
        with torch.enable_grad():
            memo = evaluate([self.loss] + [n for n in (opt.lr,) if isinstance(n, Node)], feeds)
            loss = memo[self.loss]
            leaves = [v.value for v in self.var_list]
            grads = torch.autograd.grad(loss.sum(), leaves, allow_unused=True)   # tf.gradients: d(sum loss)/dv
        lr = float(memo[opt.lr]) if isinstance(opt.lr, Node) else float(opt.lr)
        b1p, b2p = float(opt.beta1_power.value), float(opt.beta2_power.value)
        lr_t = lr * math.sqrt(1.0 - b2p) / (1.0 - b1p)
        opt.last_grads = {v.var_name: (None if g is None else g.detach().numpy().copy())
                          for v, g in zip(self.var_list, grads)}
        with torch.no_grad():
            for v, g in zip(self.var_list, grads):
                if g is None:
                    continue               # variables not connected to the loss are skipped by minimize()
                m, vv = opt.slots[v]
                m.value = (opt.beta1 * m.value + (1.0 - opt.beta1) * g).detach()
                vv.value = (opt.beta2 * vv.value + (1.0 - opt.beta2) * g * g).detach()
                v.value = (v.value - lr_t * m.value / (torch.sqrt(vv.value) + opt.epsilon)).detach().requires_grad_(True)
            opt.beta1_power.value = (opt.beta1_power.value * opt.beta1).detach()
            opt.beta2_power.value = (opt.beta2_power.value * opt.beta2).detach()
        opt.steps += 1
        for h in TRAIN_HOOKS:
            h(self, feeds, loss.detach().numpy().copy(), opt.last_grads)


class AdamOptimizer:
    OPTIMIZERS = []

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **k):
        self.lr, self.beta1, self.beta2, self.epsilon = learning_rate, float(beta1), float(beta2), float(epsilon)
        self.slots = {}
        self.steps = 0
        self.last_grads = None
        AdamOptimizer.OPTIMIZERS.append(self)

    def minimize(self, loss, var_list=None, **k):
        var_list = list(var_list) if var_list is not None else trainable_variables()
        g = _g()
        idx = sum(1 for v in g.variables if v.var_name.startswith("beta1_power"))
        sfx = "" if idx == 0 else f"_{idx}"
        for v in var_list:
            shp = tuple(v.trace.shape)
            slot_names = []
            for base in ("Adam", "Adam_1"):
                slot_names.append(g.unique_var(v.var_name + "/" + base))
            m = Variable(None, trainable=False, _full_name=slot_names[0], _init_fn=lambda s=shp: np.zeros(s))
            vv = Variable(None, trainable=False, _full_name=slot_names[1], _init_fn=lambda s=shp: np.zeros(s))
            self.slots[v] = (m, vv)
        self.beta1_power = Variable(None, trainable=False, _full_name="beta1_power" + sfx,
                                    _init_fn=lambda: np.array(self.beta1))
        self.beta2_power = Variable(None, trainable=False, _full_name="beta2_power" + sfx,
                                    _init_fn=lambda: np.array(self.beta2))
        self.var_list = var_list
        self.loss = loss
        return _TrainOp(self, convert(loss), var_list)


def placeholder(dtype=None, shape=None, name=None):
    return Placeholder(dtype, shape, name)


# ------------------------------------------------------------------------------------------------
# the ``tf`` object
# ------------------------------------------------------------------------------------------------
def _ns(**k):
    return types.SimpleNamespace(**k)


def leaky_relu(x, alpha=0.2, name=None):
    a = float(alpha)
    return Node(lambda v: torch.where(v > 0, v, a * v), (convert(x),))


def softmax(x, axis=-1, name=None):
    return Node(lambda v: torch.softmax(v, dim=axis), (convert(x),))


def expand_dims(x, axis, name=None):
    return Node(lambda v: v.unsqueeze(axis), (convert(x),))


def matmul(a, b, name=None):
    return Node(lambda u, v: torch.matmul(u, v.to(u.dtype)), (convert(a), convert(b)))


def concat(values, axis, name=None):
    ns = [convert(v) for v in values]
    return Node(lambda *v: torch.cat(v, dim=axis), ns)


def permute_dimensions(x, pattern):
    p = tuple(pattern)
    return Node(lambda v: v.permute(*p), (convert(x),))


def constant(value, dtype=None, shape=None, name=None):
    return convert(value)


def build_tf():
    tanh = _un(torch.tanh)
    log = _un(torch.log)
    v1 = _ns(
        placeholder=placeholder, variable_scope=variable_scope, trainable_variables=trainable_variables,
        global_variables=global_variables, get_collection=get_collection, GraphKeys=GraphKeys,
        global_variables_initializer=global_variables_initializer, variables_initializer=variables_initializer,
        Session=Session, py_func=py_func, reset_default_graph=reset_default_graph,
        layers=_ns(conv1d=layers_conv1d), train=_ns(AdamOptimizer=AdamOptimizer, Saver=Saver),
        log=log, math=_ns(log=log), subtract=_bi(torch.sub),
    )
    sig = _ns(stft=stft, linear_to_mel_weight_matrix=linear_to_mel_weight_matrix)
    tf = _ns(
        float32="float32", bool="bool", float64="float64", int32="int32",
        compat=_ns(v1=v1, v2=_ns(signal=sig)),
        signal=sig,
        keras=_ns(layers=_ns(SeparableConv1D=SeparableConv1D), backend=_ns(permute_dimensions=permute_dimensions)),
        nn=_ns(tanh=tanh, leaky_relu=leaky_relu, softmax=softmax, top_k=top_k, relu=_un(torch.relu)),
        math=_ns(log=log, real=_un(lambda v: v.real), imag=_un(lambda v: v.imag)),
        Graph=Graph, Variable=Variable, constant=constant, cast=cast, cond=cond, one_hot=one_hot,
        reshape=reshape, expand_dims=expand_dims, matmul=matmul, concat=concat,
        abs=_un(torch.abs), sqrt=_un(torch.sqrt), square=_un(torch.square), sign=_un(torch.sign), tanh=tanh,
        multiply=_bi(torch.mul), subtract=_bi(torch.sub), add=_bi(torch.add),
        reduce_sum=_reduce(torch.sum), reduce_mean=_reduce(torch.mean),
        custom_gradient=custom_gradient, function=lambda f: f,
    )
    return tf
