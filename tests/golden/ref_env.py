"""Import environment for running the reference (``/root/reference``) in the build container.  TEST INFRASTRUCTURE.

``install()`` registers in ``sys.modules``:
  * ``tensorflow``                     -> the lazy-graph stand-in of ``tf_shim.py`` (executes, float64);
  * ``audiolazy`` (``ZFilter``), ``spectrum`` (``lsf2poly``) -> restatements of the two third-party algorithms the
    reference's ``lpc_utilities.py`` calls (neither package is installed here, there is no lock file: the
    reference imports whatever ``pip install audiolazy spectrum`` gave its author; the published behaviour is
    audiolazy 0.6 ``ZFilter`` = direct-form linear filter with zero initial state evaluated in Python floats,
    spectrum 0.7 ``lsf2poly`` = the Kondoz sum/difference-polynomial construction) - so the LPC rows are
    **parity unpinned** beyond the reference's own call sites;
  * inert mocks for packages that only the out-of-scope evaluation code touches (librosa, pystoi, soundfile, mdct,
    tensorflow_probability).
and returns the imported reference modules.  Never used on the GPU box.
"""
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class ZFilter:
    """audiolazy.ZFilter restated: H(z) = (b0 + b1 z^-1 + ...) / (a0 + a1 z^-1 + ...); calling it filters a finite
    sequence from rest (zero state) in double precision and yields the output samples."""

    def __init__(self, numerator=None, denominator=None):
        self.numlist = [float(v) for v in (numerator if numerator is not None else [1.0])]
        self.denlist = [float(v) for v in (denominator if denominator is not None else [1.0])]

    def __truediv__(self, other):
        if isinstance(other, ZFilter):
            return ZFilter(np.convolve(self.numlist, other.denlist), np.convolve(self.denlist, other.numlist))
        return ZFilter([v / other for v in self.numlist], self.denlist)

    def __rtruediv__(self, k):
        return ZFilter([float(k) * v for v in self.denlist], self.numlist)

    def __call__(self, seq):
        x = [float(v) for v in seq]
        b, a = self.numlist, self.denlist
        y = [0.0] * len(x)
        for n in range(len(x)):
            acc = 0.0
            for k, bk in enumerate(b):
                if n - k >= 0:
                    acc += bk * x[n - k]
            for k in range(1, len(a)):
                if n - k >= 0:
                    acc -= a[k] * y[n - k]
            y[n] = acc / a[0]
        return y


def lsf2poly(lsf):
    """spectrum.lsf2poly restated (Kondoz, Digital Speech, ch. 4): roots e^{+-j w} split alternately between the
    sum and difference polynomials, known roots at z = +-1 appended, a = (P1 + Q1) / 2 without the last element."""
    lsf = np.array(lsf, dtype=np.float64)
    if lsf.max() > np.pi or lsf.min() < 0:
        raise ValueError("Line spectral frequencies must be between 0 and pi.")
    p = len(lsf)
    z = np.exp(1.0j * lsf)
    rQ, rP = z[0::2], z[1::2]
    rQ = np.concatenate((rQ, rQ.conjugate()))
    rP = np.concatenate((rP, rP.conjugate()))
    Q, P = np.poly(rQ), np.poly(rP)
    if p % 2:
        P1, Q1 = np.convolve(P, [1, 0, -1]), Q
    else:
        P1, Q1 = np.convolve(P, [1, -1]), np.convolve(Q, [1, 1])
    a = 0.5 * (P1 + Q1)
    return a[0:-1:1]


def install():
    sys.path.insert(0, HERE)
    import tf_shim
    tf = tf_shim.build_tf()
    tfmod = types.ModuleType("tensorflow")
    tfmod.__dict__.update(tf.__dict__)
    mods = {"tensorflow": tfmod}
    for name in ["tensorflow.python", "tensorflow.python.ops", "tensorflow.python.ops.math_ops",
                 "tensorflow.python.ops.random_ops", "tensorflow.python.framework",
                 "tensorflow.python.framework.dtypes", "tensorflow.python.framework.ops",
                 "tensorflow_probability", "mdct", "librosa", "pystoi", "pystoi.stoi", "soundfile"]:
        mods[name] = mock.MagicMock(name=name)
    audiolazy = types.ModuleType("audiolazy")
    audiolazy.ZFilter = ZFilter
    audiolazy.lpc = mock.MagicMock(name="audiolazy.lpc")
    audiolazy.__all__ = ["ZFilter", "lpc"]
    mods["audiolazy"] = audiolazy
    spectrum = types.ModuleType("spectrum")
    spectrum.lsf2poly = lsf2poly
    spectrum.poly2lsf = mock.MagicMock(name="spectrum.poly2lsf")
    mods["spectrum"] = spectrum
    sys.modules.update(mods)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import constants as C
        import loss_terms_and_measures as L
        import nn_core_operator as N
        import lpc_utilities as P
        import utilities as U
        import neural_speech_coding_module as M
        import cmrl as R
    finally:
        os.chdir(cwd)
    return types.SimpleNamespace(tf=tfmod, shim=tf_shim, C=C, L=L, N=N, P=P, U=U, M=M, R=R)
