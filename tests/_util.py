"""Shared helpers for the parity tests (the oracle is the checker; nsc_amd is the thing checked)."""
import ctypes as C

import numpy as np
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT

BKD = [9, 9, 100, 20, 1, 2]
RTOL = 1e-4  # north_star: fp32 conv/quantizer within 1e-4 rel


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def assert_close(a, b, tol=RTOL, what="", atol=None):
    """ELEMENTWISE bound  |a - b| <= tol * |b| + atol  (north_star: "within 1e-4 rel").  The absolute floor defaults to
    tol x rms(b): an fp32 result that went through dozens of accumulating layers carries an absolute error set by
    the typical magnitude of what was summed, so elements far below the tensor's rms cannot be held to their own
    magnitude - but they are held to 1e-4 of the rms, not (as a max-norm ratio would) of the largest element.
    Pass an explicit atol where small values matter on their own (soft-assignment probabilities: 1e-7)."""
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a64.shape == b64.shape, (what, a64.shape, b64.shape)
    assert np.all(np.isfinite(a64)), f"{what}: non-finite output"
    if atol is None:
        atol = tol * float(np.sqrt(np.mean(b64 * b64))) if b64.size else 0.0
    diff = np.abs(a64 - b64)
    err = diff - (tol * np.abs(b64) + atol)
    if b64.size and np.max(err) > 0:
        i = np.unravel_index(np.argmax(err), err.shape) if err.ndim else ()
        raise AssertionError(f"{what}: element {i}: got {a64[i]!r}, want {b64[i]!r} "
                             f"(|diff| {abs(a64[i] - b64[i]):.3e} > {tol:.1e}*|want| + {atol:.2e}); "
                             f"{int((err > 0).sum())} of {err.size} elements out of bound")
    # What the check was worth: the share of elements that passed only because of the absolute floor (their own
    # relative bound tol * |b| alone would have failed them), the largest error relative to the tensor's rms, and the largest share of
    # the bound any element used.  Returned, not asserted: the B = 128 tests print and record it.
    if not b64.size:
        return dict(floor_share=0.0, max_err_over_rms=0.0, bound_used=0.0)
    rms = float(np.sqrt(np.mean(b64 * b64)))
    return dict(floor_share=float(np.mean(diff > tol * np.abs(b64))), max_err_over_rms=float(np.max(diff) / max(rms, 1e-300)),
                bound_used=float(np.max(diff / (tol * np.abs(b64) + atol + 1e-300))))


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def synth_frames(B, seed=1234):
    """SURVEY 8d synthetic input: 0.03*N(0,1), clipped, training window."""
    rng = np.random.default_rng(seed)
    x = np.clip(0.03 * rng.standard_normal((B, 512, 1)), -1, 1) * O.training_window()[None, :, None]
    return x.astype(np.float32).astype(np.float64)


def make_store(num_codecs, strides, bins, seed=20200504, rand_bias=True, alpha=-20.0, lpc=False):
    """Oracle ParamStore with float32-representable values; optional random biases / softer alpha so that
    every gradient path is exercised (alpha=-300 saturates the softmax)."""
    ps = O.ParamStore(np.random.default_rng(seed))
    if lpc:
        ps.var("lpc_quan", "alpha", O.INIT_ALPHA)
        import json, os
        kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
        ps.var("lpc_quan", "bins", np.array(kats["lsf_bins"], np.float32).astype(np.float64))
    x0 = np.zeros((1, 512, 1))
    for i in range(num_codecs):
        O.codec_forward(x0, ps, f"scope_{i + 1}", BKD, strides[i], bins[i], 0.0, True)
    rng = np.random.default_rng(seed + 1)
    for k in ps.params:
        if rand_bias and k.endswith("/bias"):
            ps.params[k] = (0.05 * rng.standard_normal(ps.params[k].shape)).astype(np.float32).astype(np.float64)
        if alpha is not None and k.endswith("/alpha"):
            ps.params[k] = np.array(float(alpha))
    return ps
