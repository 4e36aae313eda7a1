"""Variable scopes for the op surface: replaces tf.compat.v1.variable_scope + implicit variable creation.

In the reference every op creates its variables at call time inside ``tf.compat.v1.variable_scope('scope_i')``
(neural_speech_coding_module.py:267), names auto-uniquify in creation order (conv1d, conv1d_1, ...), and trainable
lists are recovered per scope (cmrl.py:43-44).  Here a ``VariableStore`` holds torch CUDA parameters under the
same names; the first pass through a graph-building function creates them (glorot-uniform kernels, zero biases),
later passes (``store.begin_pass()``) hand the same parameters back in the same order.

Memory: CUDA variables are carved out of one flat ARENA per store (16-byte aligned, creation order), each as an independent
tensor on the arena's storage (its own version counter, no view bookkeeping).  Nothing at the surface depends on it - a variable
is an ordinary ``torch.nn.Parameter`` - but the eight tensors of a gated block then lie within one nsc_gather index range, which
is what lets ops.BlockFn build the block's kernel-ready images with one launch (the engine keeps its parameters flat for the same
reason: engine.py, ``wt_idx``).  (A variable's STORAGE is therefore the whole arena: serialise ``v.detach().clone()`` or a numpy copy,
as the trainers do, not the parameter object itself.)
"""
from __future__ import annotations

import contextlib
import math
import weakref
from collections import OrderedDict

import numpy as np
import torch

_STATE = {"store": None, "scope": ""}
_STORES = weakref.WeakSet()


def find_arena(lo, hi):
    """(store, arena tensor) of the live store whose arena holds the byte range [lo, hi), or None."""
    for st in list(_STORES):
        for a in st._arenas:
            p0 = a.data_ptr()
            if p0 <= lo and hi <= p0 + 4 * a.numel():
                return st, a
    return None


class VariableStore:
    ARENA_FLOATS = 1 << 22          # 16 MB per arena (the two-codec model: 0.95 M parameters); a full arena is followed by another

    def __init__(self, device="cuda", seed=20200504):
        self.device = torch.device(device)
        self.vars = OrderedDict()
        self.rng = np.random.default_rng(seed)
        self._counts = {}
        self._arena, self._arena_used = None, 0
        self._arenas = []
        self.pass_id = 0            # begin_pass() counts; ops.py rebuilds the kernel-ready images of the store's blocks once per pass
        self.image_sets = {}        # arena address -> ops._ImageSet
        _STORES.add(self)

    def _alloc(self, shape):
        """An uninitialised float32 tensor of `shape` inside the current arena (CUDA only; None: allocate it the ordinary way)."""
        n = int(np.prod(shape)) if len(shape) else 1
        n4 = (n + 3) // 4 * 4
        if self.device.type != "cuda" or n4 > self.ARENA_FLOATS:
            return None
        if self._arena is None or self._arena_used + n4 > self._arena.numel():
            self._arena, self._arena_used = torch.zeros(self.ARENA_FLOATS, dtype=torch.float32, device=self.device), 0
            self._arenas.append(self._arena)
        strides, acc = [], 1
        for d in reversed(shape):
            strides.append(acc)
            acc *= int(d)
        t = torch.empty(0, dtype=torch.float32, device=self.device)
        t.set_(self._arena.untyped_storage(), self._arena_used, tuple(int(d) for d in shape), tuple(reversed(strides)))
        self._arena_used += n4
        return t

    def begin_pass(self):
        """Start re-tracing the graph: layer-name counters restart, existing variables are reused.  This is also the point after
        which the pass READS the parameters: whatever the kernels derive from them (flipped kernels, kernel-ready images) is rebuilt
        on first use after it, once for the whole store (ops._ImageSet) - change parameters between passes, not inside one."""
        self._counts = {}
        self.pass_id += 1

    def uniq(self, base):
        scope = _STATE["scope"]
        n = self._counts.get((scope, base), 0)
        self._counts[(scope, base)] = n + 1
        name = base if n == 0 else f"{base}_{n}"
        return f"{scope}/{name}" if scope else name

    def get(self, name, shape, init):
        v = self.vars.get(name)
        if v is None:
            val = torch.from_numpy(np.ascontiguousarray(np.asarray(init(shape), np.float32))).reshape(tuple(shape))
            t = self._alloc(tuple(shape))
            if t is None:
                t = val.to(self.device)
            else:
                t.copy_(val)
            v = torch.nn.Parameter(t)
            self.vars[name] = v
        elif tuple(v.shape) != tuple(shape):
            raise ValueError(f"variable {name} exists with shape {tuple(v.shape)}, requested {tuple(shape)}")
        return v

    def glorot(self, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return lambda shape: self.rng.uniform(-lim, lim, size=shape)

    def trainable_variables(self, scope=None):
        """tf.compat.v1.get_collection(TRAINABLE_VARIABLES, scope=...) in creation order."""
        return [v for k, v in self.vars.items() if scope is None or k.startswith(scope + "/")]

    def named(self, scope=None):
        return OrderedDict((k, v) for k, v in self.vars.items() if scope is None or k.startswith(scope + "/"))


def current_store() -> VariableStore:
    if _STATE["store"] is None:
        _STATE["store"] = VariableStore()
    return _STATE["store"]


def set_store(store):
    _STATE["store"] = store


@contextlib.contextmanager
def variable_scope(name):
    old = _STATE["scope"]
    _STATE["scope"] = f"{old}/{name}" if old else name
    try:
        yield
    finally:
        _STATE["scope"] = old
