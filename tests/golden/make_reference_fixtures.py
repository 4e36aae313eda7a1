#!/usr/bin/env python3
"""Generate tests/golden/reference_kats.json by IMPORTING the reference (read-only, /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_reference_fixtures.py

TensorFlow and the audio third-party packages are absent here, so inert stub modules stand in for
them; only the reference's *pure NumPy* helpers are executed for real, and the encoder/decoder
builders are run against a *recording* ``tf`` stub to capture the layer topology.  The output file
holds data only (inputs + expected outputs) - no reference source text.
"""
import json
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def install_stubs(trace):
    class Shape(tuple):
        def as_list(self):
            return list(self)

    class T:  # symbolic tensor with a static shape
        def __init__(self, shape):
            self.shape = Shape(shape)

        def get_shape(self):
            return self.shape

        def __add__(self, o):
            s = self.shape if len(self.shape) >= len(getattr(o, "shape", ())) else o.shape
            so = getattr(o, "shape", ())
            out = tuple(max(a, b) if (a is not None and b is not None) else None
                        for a, b in zip(self.shape, so)) if len(so) == len(self.shape) else s
            trace.append(["add", list(self.shape), list(so), list(out)])
            return T(out)

        __radd__ = __add__

        def __getitem__(self, idx):
            return T(self.shape[:1] + self.shape[1:])

    def same_len(T_in, s):
        return -(-T_in // s)

    tf = mock.MagicMock(name="tf")

    def layers_conv1d(inputs, filters, padding, kernel_size, activation, dilation_rate, strides, data_format):
        B, Tn, C = inputs.shape
        out = (B, same_len(Tn, strides), filters)
        trace.append(["conv1d", list(inputs.shape), int(filters), int(kernel_size), int(dilation_rate),
                      int(strides), "tanh" if activation is tf.nn.tanh else ("none" if activation is None else "?"),
                      list(out)])
        return T(out)

    tf.compat.v1.layers.conv1d = layers_conv1d

    def sepconv(filters, padding, kernel_size, activation, dilation_rate, strides, data_format):
        def call(inputs):
            B, Tn, C = inputs.shape
            out = (B, same_len(Tn, strides), filters)
            trace.append(["separable_conv1d", list(inputs.shape), int(filters), int(kernel_size),
                          "none" if activation is None else "?", list(out)])
            return T(out)
        return call

    tf.keras.layers.SeparableConv1D = sepconv

    def lrelu(x):
        trace.append(["leaky_relu", list(x.shape)])
        return T(x.shape)

    tf.nn.leaky_relu = lrelu

    def multiply(a, b):
        trace.append(["multiply", list(a.shape)])
        return T(a.shape)

    tf.multiply = multiply

    def reshape(x, shape):
        return T(tuple(None if s == -1 else int(s) for s in shape))

    tf.reshape = reshape

    def permute(x, perm):
        return T(tuple(x.shape[i] for i in perm))

    tf.keras.backend.permute_dimensions = permute

    mods = {"tensorflow": tf, "tensorflow.python": tf.python, "tensorflow.python.ops": tf.python.ops,
            "tensorflow.python.ops.math_ops": tf.python.ops.math_ops,
            "tensorflow.python.ops.random_ops": tf.python.ops.random_ops,
            "tensorflow.python.framework": tf.python.framework,
            "tensorflow.python.framework.dtypes": tf.python.framework.dtypes}
    for name in ["tensorflow_probability", "mdct", "librosa", "spectrum", "pystoi", "pystoi.stoi",
                 "soundfile", "scipy.io.wavfile"]:
        mods[name] = mock.MagicMock(name=name)
    audiolazy = types.ModuleType("audiolazy")
    audiolazy.ZFilter = mock.MagicMock()
    audiolazy.lpc = mock.MagicMock()
    audiolazy.z = mock.MagicMock()
    audiolazy.lazy_lpc = mock.MagicMock()
    mods["audiolazy"] = audiolazy
    sys.modules.update(mods)
    return tf, T


def main():
    trace = []
    tf, T = install_stubs(trace)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import utilities as U          # noqa: E402  (reference module)
        import loss_terms_and_measures as L  # noqa: E402
        import constants as C          # noqa: E402
        import cmrl as M               # noqa: E402
    finally:
        os.chdir(cwd)

    kats = {}
    # ---- framing / windows (utilities.py:7-39) ----
    rng = np.random.default_rng(7)
    utt = rng.standard_normal(2000)
    kats["utt_seed"] = 7
    kats["utt_len"] = 2000
    kats["seg_post"] = U.utterance_to_segment(utt, True).tolist()
    kats["seg_win"] = U.utterance_to_segment(utt, False).tolist()
    kats["frame_counts"] = {str(n): int(U.utterance_to_segment(np.zeros(n), True).shape[0])
                            for n in [512, 513, 992, 993, 1473, 16000, 48000]}
    ones = np.ones(512)
    kats["hann_first"] = U.hann_process(ones, 0, 3).tolist()
    kats["hann_mid"] = U.hann_process(ones, 1, 3).tolist()
    kats["hann_last"] = U.hann_process(ones, 2, 3).tolist()
    # ---- scalar helpers (loss_terms_and_measures.py:63-74, 36-49, 270-277) ----
    kats["entropy_to_bitrate"] = [[2.2, 2, float(L.entropy_to_bitrate(2.2, 2))],
                                  [2.2, 4, float(L.entropy_to_bitrate(2.2, 4))],
                                  [1.5, 2, float(L.entropy_to_bitrate(1.5, 2))]]
    kats["bitrate_to_entropy"] = [[9, 2, float(L.bitrate_to_entropy(9, 2))],
                                  [24, 4, float(L.bitrate_to_entropy(24, 4))]]
    r0 = np.random.default_rng(0)
    a = r0.standard_normal(1000)
    b = a + 0.1 * r0.standard_normal(1000)
    kats["snr_seed0"] = float(L.snr(a, b)[1])
    kats["si_snr_seed0"] = float(L.si_snr(b, a))
    # ---- constants (constants.py) ----
    kats["lsf_bins"] = [float(v) for v in C.lpc_coeff_lsf_bins]
    kats["constants"] = dict(init_alpha=C.init_alpha, beta_boundary=C.beta_boundary, frame_length=C.frame_length,
                             overlap_each_side=C.overlap_each_side, sample_rate=C.sample_rate,
                             max_amp_tr=C.max_amp_tr, resnet_type=C.resnet_type,
                             is_pure_time_domain=C.is_pure_time_domain)
    # ---- topology trace of the builders (neural_speech_coding_module.py:152-260) ----
    topo = {}
    for key, strides in (("2", [2]), ("2_2", [2, 2])):
        obj = M.CMRL.__new__(M.CMRL)
        obj._bottleneck_kernel_and_dilation = [9, 9, 100, 20, 1, 2]
        import io
        import contextlib
        del trace[:]
        with contextlib.redirect_stdout(io.StringIO()):
            _, code = obj._the_encoder_in_each_module(T((128, 512, 1)), strides, True)
            n_enc = len(trace)
            _, out = obj._the_decoder_in_each_module(T(code.shape), strides, True)
        topo[key] = dict(encoder=list(trace[:n_enc]), decoder=list(trace[n_enc:]),
                         code_shape=list(code.shape), out_shape=list(out.shape))
    kats["topology"] = topo
    with open(OUT, "w") as f:
        json.dump(kats, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for key in topo:
        ops = [t[0] for t in topo[key]["encoder"] + topo[key]["decoder"]]
        print(key, {o: ops.count(o) for o in sorted(set(ops))}, topo[key]["code_shape"], topo[key]["out_shape"])


if __name__ == "__main__":
    main()
