"""Variable scopes for the op surface: replaces tf.compat.v1.variable_scope + implicit variable creation.

In the reference every op creates its variables at call time inside ``tf.compat.v1.variable_scope('scope_i')``
(neural_speech_coding_module.py:267), names auto-uniquify in creation order (conv1d, conv1d_1, ...), and trainable
lists are recovered per scope (cmrl.py:43-44).  Here a ``VariableStore`` holds torch CUDA parameters under the
same names; the first pass through a graph-building function creates them (glorot-uniform kernels, zero biases),
later passes (``store.begin_pass()``) hand the same parameters back in the same order.
"""
from __future__ import annotations

import contextlib
import math
from collections import OrderedDict

import numpy as np
import torch

_STATE = {"store": None, "scope": ""}


class VariableStore:
    def __init__(self, device="cuda", seed=20200504):
        self.device = torch.device(device)
        self.vars = OrderedDict()
        self.rng = np.random.default_rng(seed)
        self._counts = {}

    def begin_pass(self):
        """Start re-tracing the graph: layer-name counters restart, existing variables are reused."""
        self._counts = {}

    def uniq(self, base):
        scope = _STATE["scope"]
        n = self._counts.get((scope, base), 0)
        self._counts[(scope, base)] = n + 1
        name = base if n == 0 else f"{base}_{n}"
        return f"{scope}/{name}" if scope else name

    def get(self, name, shape, init):
        v = self.vars.get(name)
        if v is None:
            v = torch.nn.Parameter(torch.tensor(np.asarray(init(shape), np.float32), device=self.device))
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
