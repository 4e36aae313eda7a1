"""CPU ORACLE (float64 NumPy) for the NSC/CMRL hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path (``nsc_amd``)
must never import anything from ``oracle/``.

PARITY STATUS: pinned to outputs of the reference's OWN code run in the build container
(tests/golden/make_reference_exec.py -> tests/golden/reference_exec.npz, checked by
tests/test_reference_exec.py to <= 1e-9): the quantizer, the gated block, the four losses,
the codec graphs ([2] and [2,2], plain and _lpc), the 2- and 4-codec cascades, every phase's
optimised [B] loss vector, first-step gradients, multi-step Adam trajectories / checkpoints,
the tau controllers, the validation entropies, utterance-level inference and the LPC
utilities; plus the pure-NumPy helpers (tests/golden/make_reference_fixtures.py).
The reference ships no tests or golden vectors and TensorFlow is not installable here, so
that run executes the reference's Python on a lazy-graph stand-in whose PRIMITIVES
(conv1d / SeparableConv1D with SAME padding, leaky_relu, softmax, top_k, one_hot, stft,
the mel matrix, TF1 Adam, sum-of-vector-loss gradients) are restatements of TensorFlow's
documented semantics ([TF-semantics] tags below): those primitives, and audiolazy.ZFilter /
spectrum.lsf2poly behind the LPC rows, remain **parity unpinned**; every composition above
them is the reference's own code.

Layout conventions follow the reference: activations channels_last ``[B, T, C]``,
conv kernels ``[K, Cin, Cout]`` (TF), everything float64 here.

Each function cites the reference file:line it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# constants.py:5,8,25,26
INIT_ALPHA = -300.0
BETA_BOUNDARY = 1.0
FRAME_LENGTH = 512
OVERLAP_EACH_SIDE = 32
SAMPLE_RATE = 16000
LRELU_ALPHA = 0.2  # tf.nn.leaky_relu default [TF-semantics], nn_core_operator.py:30


# --------------------------------------------------------------------------------------
# SAME padding [TF-semantics]  
# --------------------------------------------------------------------------------------
def same_pad(T, k, dil=1, stride=1):
    """TF 'SAME': T_out = ceil(T/s); pad = max((T_out-1)*s + (k-1)*d + 1 - T, 0); padL = pad//2."""
    t_out = -(-T // stride)
    pad = max((t_out - 1) * stride + (k - 1) * dil + 1 - T, 0)
    return t_out, pad // 2, pad - pad // 2


def leaky_relu(x):
    """nn_core_operator.py:24-31 (activation_func) -> tf.nn.leaky_relu, alpha 0.2."""
    return np.where(x > 0, x, LRELU_ALPHA * x)


def _act(x, activation):
    if activation is None or activation == "none":
        return x
    if activation == "tanh":
        return np.tanh(x)
    if activation == "lrelu":
        return leaky_relu(x)
    raise ValueError(activation)


# --------------------------------------------------------------------------------------
# conv ops (nn_core_operator.py:6-21)
# --------------------------------------------------------------------------------------
def conv1d(x, W, b, dilation_rate=1, strides=1, activation="tanh"):
    """nn_core_operator.py:6-14.  y[b,t,o] = act(bias[o] + sum_{k,i} x[b, t*s + k*d - padL, i] W[k,i,o]).

    Cross-correlation (no kernel flip), zero padding, channels_last.  Default activation of the
    reference wrapper is tanh (callers always pass it explicitly).
    """
    x = np.asarray(x, np.float64)
    W = np.asarray(W, np.float64)
    B, T, Cin = x.shape
    K, Cin2, Cout = W.shape
    assert Cin == Cin2
    t_out, pad_l, pad_r = same_pad(T, K, dilation_rate, strides)
    xp = np.zeros((B, T + pad_l + pad_r, Cin))
    xp[:, pad_l:pad_l + T] = x
    y = np.zeros((B, t_out, Cout))
    for k in range(K):
        start = k * dilation_rate
        seg = xp[:, start:start + (t_out - 1) * strides + 1:strides, :]
        y += seg @ W[k]
    if b is not None:
        y = y + np.asarray(b, np.float64)
    return _act(y, activation)


def conv1d_depth(x, Wd, Wp, b, activation=None):
    """nn_core_operator.py:17-21: Keras SeparableConv1D [TF-semantics]: depthwise [K,C,1] (multiplier 1),
    pointwise [1,C,Cout], one bias after the pointwise conv, SAME, stride 1, dilation 1."""
    x = np.asarray(x, np.float64)
    B, T, C = x.shape
    K = Wd.shape[0]
    assert Wd.shape == (K, C, 1)
    t_out, pad_l, pad_r = same_pad(T, K)
    xp = np.zeros((B, T + pad_l + pad_r, C))
    xp[:, pad_l:pad_l + T] = x
    dw = np.zeros((B, T, C))
    for k in range(K):
        dw += xp[:, k:k + T, :] * Wd[k, :, 0]
    y = dw @ np.asarray(Wp, np.float64)[0] + np.asarray(b, np.float64)
    return _act(y, activation)


def subpixel_shuffle(x, s=2):
    """neural_speech_coding_module.py:158-167: out[b, t*s+j, c] = in[b, t, c*s+j]."""
    B, T, C = x.shape
    r = x.reshape(B, T, C // s, s)
    r = np.transpose(r, (0, 1, 3, 2))
    return r.reshape(B, T * s, C // s)


# --------------------------------------------------------------------------------------
# Parameter store: replaces TF variable scopes (creation order == TF trainable_variables order)
# --------------------------------------------------------------------------------------
def name_seeded_uniform(name, shape, lim, seed=20200504):
    """U(-lim, lim) from a generator seeded by (seed, crc32(name)); float32-representable."""
    import zlib
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    return rng.uniform(-lim, lim, size=shape).astype(np.float32).astype(np.float64)


class ParamStore:
    """Ordered name->array store with TF1-style auto-uniquified layer names
    (conv1d, conv1d_1, ...; separable_conv1d, ...) inside a scope (nsc_module:267)."""

    def __init__(self, rng=None, name_seeded=False):
        self.params = OrderedDict()
        self._counts = {}
        self.rng = rng if rng is not None else np.random.default_rng(20200504)
        self.replay = False  # when True, get() returns existing params in creation order
        # name_seeded: every kernel's initial value is a function of its TF variable name (the recipe the
        # reference-executed fixtures were generated with: tests/golden/tf_shim.py::name_seeded_uniform)
        self.name_seeded = name_seeded

    def _uniq(self, scope, base):
        key = (scope, base)
        n = self._counts.get(key, 0)
        self._counts[key] = n + 1
        return f"{scope}/{base}" if n == 0 else f"{scope}/{base}_{n}"

    def begin_replay(self):
        self._counts = {}
        self.replay = True

    def glorot(self, shape, fan_in, fan_out, name=None):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        if self.name_seeded:
            return name_seeded_uniform(name, shape, lim)
        # float32-representable values so every implementation starts from identical weights
        return self.rng.uniform(-lim, lim, size=shape).astype(np.float32).astype(np.float64)

    def conv(self, scope, K, Cin, Cout):
        name = self._uniq(scope, "conv1d")
        if not self.replay:
            # glorot_uniform fans for a [K,Cin,Cout] kernel: K*Cin / K*Cout [TF-semantics]
            self.params[name + "/kernel"] = self.glorot((K, Cin, Cout), K * Cin, K * Cout, name + "/kernel")
            self.params[name + "/bias"] = np.zeros(Cout)
        return self.params[name + "/kernel"], self.params[name + "/bias"]

    def sepconv(self, scope, K, C, Cout):
        name = self._uniq(scope, "separable_conv1d")
        if not self.replay:
            # Keras: depthwise [K,C,1] -> fan_in=K*C, fan_out=K*1; pointwise [1,C,Cout] -> C / Cout
            self.params[name + "/depthwise_kernel"] = self.glorot((K, C, 1), K * C, K, name + "/depthwise_kernel")
            self.params[name + "/pointwise_kernel"] = self.glorot((1, C, Cout), C, Cout, name + "/pointwise_kernel")
            self.params[name + "/bias"] = np.zeros(Cout)
        return (self.params[name + "/depthwise_kernel"], self.params[name + "/pointwise_kernel"],
                self.params[name + "/bias"])

    def var(self, scope, name, value):
        full = f"{scope}/{name}"
        if not self.replay:
            self.params[full] = np.array(value, dtype=np.float64)
        return self.params[full]

    def scope_names(self, scope):
        return [k for k in self.params if k.startswith(scope + "/")]


# --------------------------------------------------------------------------------------
# blocks and codec (nn_core_operator.py:82-112, neural_speech_coding_module.py:152-295)
# --------------------------------------------------------------------------------------
def gated_bottleneck(x, ps, scope, wide_layer, narrow_layer, non_dilated_neck_kernel_size=9,
                     dilation_rate=1, is_last_flat=False, tape=None):
    """nn_core_operator.py:82-112.  1x1 -> lrelu -> {conv15_d, conv15_d+tanh} -> mul -> conv9 -> +x -> lrelu.
    dilated kernel size is hard-coded 15 (:92,:97).  Residual add broadcasts when Cin == 1."""
    Cin = x.shape[-1]
    W1, b1 = ps.conv(scope, 1, Cin, narrow_layer)
    h = leaky_relu(conv1d(x, W1, b1, activation=None))
    Wl, bl = ps.conv(scope, 15, narrow_layer, narrow_layer)
    left = conv1d(h, Wl, bl, dilation_rate=dilation_rate, activation=None)
    Wr, br = ps.conv(scope, 15, narrow_layer, narrow_layer)
    right = conv1d(h, Wr, br, dilation_rate=dilation_rate, activation="tanh")
    g = left * right
    W9, b9 = ps.conv(scope, non_dilated_neck_kernel_size, narrow_layer, wide_layer)
    y = conv1d(g, W9, b9, activation=None) + x
    out = y if is_last_flat else leaky_relu(y)
    if tape is not None:
        tape.append(("gated_bottleneck", dict(h=h, left=left, right=right, g=g, out=out)))
    return out


def stack_bottleneck_blocks(x, ps, scope, bkd, tape=None):
    """neural_speech_coding_module.py:183-217 (resnet_type == 'gln')."""
    wide = bkd[2] if x.shape[-1] == 1 else x.shape[-1]
    n = len(bkd) - 4
    for i in range(n):
        x = gated_bottleneck(x, ps, scope, wide_layer=wide, narrow_layer=bkd[3],
                             non_dilated_neck_kernel_size=bkd[1], dilation_rate=bkd[i + 4],
                             is_last_flat=(i == n - 1), tape=tape)
    return x


def encoder(x, ps, scope, bkd, strides, tape=None):
    """neural_speech_coding_module.py:219-237."""
    W, b = ps.conv(scope, 55, x.shape[-1], bkd[2])
    h = leaky_relu(conv1d(x, W, b, activation=None))
    for s in strides:
        h = stack_bottleneck_blocks(h, ps, scope, bkd, tape)
        W, b = ps.conv(scope, 9, h.shape[-1], bkd[2])  # _down_sampling_mod :152-156
        h = leaky_relu(conv1d(h, W, b, strides=s, activation=None))
    h = stack_bottleneck_blocks(h, ps, scope, bkd, tape)
    W, b = ps.conv(scope, 55, h.shape[-1], 1)
    return conv1d(h, W, b, activation="tanh")


def decoder(code, ps, scope, bkd, strides, tape=None):
    """neural_speech_coding_module.py:239-260."""
    h = code
    for s in strides:
        h = stack_bottleneck_blocks(h, ps, scope, bkd, tape)
        C = h.shape[-1]
        Wd, Wp, b = ps.sepconv(scope, 9, C, C)  # _up_sampling_mod :169-181
        h = subpixel_shuffle(leaky_relu(conv1d_depth(h, Wd, Wp, b, activation=None)), s)
    h = stack_bottleneck_blocks(h, ps, scope, bkd, tape)
    W, b = ps.conv(scope, 55, h.shape[-1], 1)
    return conv1d(h, W, b, activation=None)


def softmax_lastaxis(z):
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def scalar_softmax_quantization(floating_code, alpha, bins, is_quan_on, the_share):
    """nn_core_operator.py:140-164.  Returns (soft p [B,L,nb], bit_code [B,L,1]).
    First output is ALWAYS the soft assignment (:147,:164).  Hard = one_hot(argmax p), lowest index
    on ties (tf.nn.top_k) [TF-semantics]."""
    c = np.asarray(floating_code, np.float64)
    bins = np.asarray(bins, np.float64)
    dist = np.abs(c - bins.reshape(1, 1, -1))
    p = softmax_lastaxis(float(alpha) * dist)
    if the_share:
        sel = p
    else:
        idx = np.argmax(p, axis=-1)  # np.argmax returns the lowest index among ties
        sel = np.eye(len(bins))[idx]
    q = (sel @ bins)[..., None]
    out = (1.0 - is_quan_on) * c + is_quan_on * q
    return p, out


def codec_forward(x, ps, scope, bkd, strides, num_bins, is_quan_on, the_share, tape=None):
    """neural_speech_coding_module.py:262-295 (computational_graph_end2end_quan_on).
    Returns dict with soft assignment p, floating code, quantized code, decoded [B,512]."""
    alpha = ps.var(scope, "alpha", INIT_ALPHA)
    # tf.Variable(np.linspace(...), dtype=tf.float32): the initial bins are float32-rounded (nsc_module:269)
    bins = ps.var(scope, "bins", np.linspace(-BETA_BOUNDARY, BETA_BOUNDARY, num_bins).astype(np.float32))
    code = encoder(x, ps, scope, bkd, strides, tape)
    p, qcode = scalar_softmax_quantization(code, alpha, bins, is_quan_on, the_share)
    dec = decoder(qcode, ps, scope, bkd, strides, tape)
    return dict(p=p, floating_code=code, code=qcode, decoded=dec[:, :, 0], alpha=alpha, bins=bins)


# --------------------------------------------------------------------------------------
# losses (loss_terms_and_measures.py)
# --------------------------------------------------------------------------------------
def mse_loss(decoded, original):
    """loss_terms_and_measures.py:77-79: sqrt(mean_t (d-o)^2 + 1e-7) -> [B]."""
    return np.sqrt(np.mean((decoded - original) ** 2, axis=-1) + 1e-7)


def hertz_to_mel(f, dtype=np.float64):
    return (dtype(1127.0) * np.log1p(np.asarray(f, dtype) / dtype(700.0))).astype(dtype)


def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins=257, sample_rate=16000,
                                lower_edge_hertz=0.0, upper_edge_hertz=8000.0, dtype=np.float64):
    """tf.signal.linear_to_mel_weight_matrix [TF-semantics] (loss_terms_and_measures.py:138):
    HTK mel scale, DC bin zeroed, triangles built in the mel domain.
    dtype: the precision of the whole computation.  The shim / oracle / product take float64 then cast the result to float32
    (TensorFlow >= 2.1 computes in float64 internally); TensorFlow 2.0 - the README badge of the reference - computes every step in
    the requested dtype, float32.  tests/test_mel_precision.py bounds what that choice does to mfcc_loss and its gradient."""
    nyquist = dtype(sample_rate / 2.0)
    lin = np.linspace(dtype(0.0), nyquist, num_spectrogram_bins, dtype=dtype)[1:]
    spec_mel = hertz_to_mel(lin, dtype)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz, dtype), hertz_to_mel(upper_edge_hertz, dtype), num_mel_bins + 2, dtype=dtype)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(dtype(0.0), np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[1, 0], [0, 0]]).astype(dtype)


MEL_BANKS = (8, 16, 32, 128)  # loss_terms_and_measures.py:133


def mel_matrix_cat(dtype=np.float64):
    """The 4 banks concatenated to [257, 184]; cast through float32 like TF's returned matrix (dtype: see above)."""
    return np.concatenate([linear_to_mel_weight_matrix(n, dtype=dtype) for n in MEL_BANKS], axis=1) \
        .astype(np.float32).astype(np.float64)


def tf_stft(sig):
    """loss_terms_and_measures.py:178-183: window_fn=None, frame_len=step=fft=512 => bare rFFT-512 per row.
    mag = sqrt(re^2 + im^2 + 1e-7)."""
    sig = np.asarray(sig, np.float64).reshape(-1, FRAME_LENGTH)
    st = np.fft.rfft(sig, n=FRAME_LENGTH, axis=-1)
    mag = np.sqrt(st.real ** 2 + st.imag ** 2 + 1e-7)
    return st, mag


def rfft512_direct(sig):
    """Direct O(N^2) DFT used to pin tf_stft's use of np.fft (X_k = sum_n x_n e^{-2 pi i k n / N})."""
    sig = np.asarray(sig, np.float64).reshape(-1, FRAME_LENGTH)
    n = np.arange(FRAME_LENGTH)
    k = np.arange(FRAME_LENGTH // 2 + 1)
    ang = -2.0 * np.pi * np.outer(n, k) / FRAME_LENGTH
    return sig @ np.cos(ang) + 1j * (sig @ np.sin(ang))


def mfcc_loss(decoded, original):
    """loss_terms_and_measures.py:151-175 (+ mfcc_transform :130-148).  PSD = mag^2/512; 4 mel banks;
    log(.+1e-7); per-bank sqrt(mean (d-o)^2 + 1e-7); mean over the 4 banks -> [B].
    The shape[0]==128 branch (:169-170) is dead (None batch dim) and is not restated."""
    _, dm = tf_stft(decoded)
    _, om = tf_stft(original)
    dpsd = dm ** 2 / FRAME_LENGTH
    opsd = om ** 2 / FRAME_LENGTH
    M = mel_matrix_cat()
    dl = np.log(dpsd @ M + 1e-7)
    ol = np.log(opsd @ M + 1e-7)
    out = []
    off = 0
    for n in MEL_BANKS:
        out.append(np.sqrt(np.mean((dl[:, off:off + n] - ol[:, off:off + n]) ** 2, axis=-1) + 1e-7))
        off += n
    return np.mean(np.stack(out, -1), axis=-1)


def quan_loss(p):
    """loss_terms_and_measures.py:257-259: mean_l sum_k sqrt(p + 1e-20) -> [B]."""
    return np.mean(np.sum(np.sqrt(p + 1e-20), axis=-1), axis=-1)


def entropy_coding_loss(p, hist_extra=None):
    """loss_terms_and_measures.py:262-267: histogram over the WHOLE batch, scalar.
    hist_extra lets a caller add other ranks' histograms (data-parallel global batch)."""
    h = p.reshape(-1, p.shape[-1]).sum(axis=0)
    if hist_extra is not None:
        h = h + hist_extra
    h = h / h.sum()
    return -np.sum(h * np.log(h + 1e-7) / np.log(2.0))


def entropy_to_bitrate(total_entropy, the_strides):
    """loss_terms_and_measures.py:63-67."""
    code_len_val = 128 if the_strides == 4 else 256
    return ((SAMPLE_RATE / 1024.0) / (FRAME_LENGTH - OVERLAP_EACH_SIDE)) * code_len_val * total_entropy


def bitrate_to_entropy(bitrate, the_strides):
    """loss_terms_and_measures.py:70-74 (kept verbatim, including its operator-precedence quirk)."""
    pre = (FRAME_LENGTH / the_strides) * (float(FRAME_LENGTH / the_strides) / FRAME_LENGTH)
    entropy = (bitrate / pre * SAMPLE_RATE)
    entropy *= (FRAME_LENGTH - OVERLAP_EACH_SIDE / float(FRAME_LENGTH))
    return entropy


def snr(ori_sig, dec_sig):
    """loss_terms_and_measures.py:270-277."""
    n = min(len(ori_sig), len(dec_sig))
    o, d = ori_sig[:n], dec_sig[:n]
    return n, 10 * np.log10(np.sum(o ** 2) / (np.sum((o - d) ** 2) + 1e-20) + 1e-20), o, d


def si_snr(x, s):
    """loss_terms_and_measures.py:36-49."""
    x_zm = x - np.mean(x)
    s_zm = s - np.mean(s)
    t = (np.inner(x_zm, s_zm) / np.linalg.norm(s_zm, 2) ** 2) * s_zm
    n = x_zm - t
    return 20 * np.log10(np.linalg.norm(t, 2) / np.linalg.norm(n, 2))


# --------------------------------------------------------------------------------------
# framing helpers (utilities.py:7-39) -- pinned against the reference import
# --------------------------------------------------------------------------------------
def training_window():
    """utilities.py:28-30: hanning(63)[:32] | ones(448) | hanning(63)[31:]."""
    o = OVERLAP_EACH_SIDE
    return np.concatenate([np.hanning(2 * o - 1)[:o], np.ones(FRAME_LENGTH - 2 * o), np.hanning(2 * o - 1)[o - 1:]])


def utterance_to_segment(utterance, post_window=False):
    """utilities.py:25-39: hop 480, frame i starts at 480*i for i in range(0, len-512, 480)."""
    hop = FRAME_LENGTH - OVERLAP_EACH_SIDE
    starts = list(range(0, len(utterance) - FRAME_LENGTH, hop))
    ret = np.empty((len(starts), FRAME_LENGTH))
    win = np.ones(FRAME_LENGTH) if post_window else training_window()
    for j, i in enumerate(starts):
        ret[j] = utterance[i:i + FRAME_LENGTH] * win
    return ret


def hann_process(seg, seg_ind, seg_amount):
    """utilities.py:7-22: three window variants (first / middle / last)."""
    o = OVERLAP_EACH_SIDE
    mid = FRAME_LENGTH - 2 * o
    if seg_ind == 0:
        w = np.concatenate([np.ones(o), np.ones(mid), np.hanning(2 * o)[o:]])
    elif seg_ind == seg_amount - 1:
        w = np.concatenate([np.hanning(2 * o)[:o], np.ones(mid), np.ones(o)])
    else:
        w = training_window()
    return seg * w


def overlap_add(frames):
    """cmrl.py:595-597 / nsc_module eval loops: Hann-window each decoded frame, add at hop 480."""
    n = frames.shape[0]
    hop = FRAME_LENGTH - OVERLAP_EACH_SIDE
    out = np.zeros(hop * (n - 1) + FRAME_LENGTH if n else 0)
    for i in range(n):
        out[i * hop:i * hop + FRAME_LENGTH] += hann_process(frames[i], i, n)
    return out


# --------------------------------------------------------------------------------------
# optimizer (tf.compat.v1.train.AdamOptimizer [TF-semantics], nsc_module:922-925)
# --------------------------------------------------------------------------------------
def adam_tf1_step(theta, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """TF1 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t * m / (sqrt(v) + eps)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------------------
# cascade (cmrl.py:22-135, 295-390, 392-511, 513-543) and loss assembly (nsc_module:908-918)
# --------------------------------------------------------------------------------------
def cascade_forward(x, ps, bkd, strides_per_codec, bins_per_codec, is_quan_on, the_share, res_scalar=1.0,
                    scale_first=False):
    """codec i input = res_scalar * (x - sum_{j<i} yhat_j); yhat_i = dec_i / res_scalar; out = sum_i yhat_i.
    cmrl.py:49-58, 77-84, 94 (followers) / :308-327 (finetune).  NOTE the reference applies res_scalar
    only to codecs i >= 1 in the time-domain path (codec 0 gets x un-scaled, :30-37) and to every codec
    in the LPC path (:167-178) -> ``scale_first=True`` there."""
    outs = []
    yhat = []
    for i, (st, nb) in enumerate(zip(strides_per_codec, bins_per_codec)):
        if i == 0 and not scale_first:
            xin = x
            o = codec_forward(xin, ps, f"scope_{i + 1}", bkd, st, nb, is_quan_on, the_share)
            y = o["decoded"]
        elif i == 0:
            o = codec_forward(res_scalar * x, ps, f"scope_{i + 1}", bkd, st, nb, is_quan_on, the_share)
            y = o["decoded"] / res_scalar
        else:
            xin = res_scalar * (x - np.sum(yhat, axis=0)[..., None])
            o = codec_forward(xin, ps, f"scope_{i + 1}", bkd, st, nb, is_quan_on, the_share)
            y = o["decoded"] / res_scalar
        yhat.append(y)
        outs.append(o)
    return outs, np.sum(yhat, axis=0)


def loss_terms(decoded, target, p_list):
    """time, freq, per-codec quan [B] and entropy scalars."""
    return dict(time=mse_loss(decoded, target), freq=mfcc_loss(decoded, target),
                quan=[quan_loss(p) for p in p_list], ent=[entropy_coding_loss(p) for p in p_list])


def phase_loss(terms, coeff, tau, mode, lpc_terms=None, code_lens=None):
    """The [B] loss VECTOR each phase hands to AdamOptimizer.minimize (scalars broadcast into it, exactly as the
    reference's Python builds it).  terms = loss_terms(...); lpc_terms = dict(quan=[B], ent=scalar) of the LSF
    quantizer; code_lens = (16, L) for the one_ae_lpc blend.
      'no_quan'      c0*time + c1*freq                                              nsc_module:914, cmrl.py:99
      'quan_last'    ... + c2*quan[-1] + tau*ent[-1]                                nsc_module:915-918, cmrl.py:101-104
      'finetune'     ... + c2*SUM_{i,b} quan_i[b] (tf.reduce_sum of the LIST: a scalar, cmrl.py:355)
                         + sum_i tau[i]*ent_i                                       cmrl.py:361-365
      'one_ae_lpc'   ... + c2*(quan_lpc*16/(16+L) + quan*L/(16+L)) + tau*(16/(16+L)*ent_lpc + L/(16+L)*ent)
                                                                                     nsc_module:1032-1050
      'finetune_lpc' ... + c2*(quan_lpc + sum_i quan_i)   (no entropy term)         cmrl.py:463-485
    """
    base = coeff[0] * terms["time"] + coeff[1] * terms["freq"]
    tau = np.ravel(np.asarray(tau, np.float64))
    if mode == "no_quan":
        return base
    if mode == "quan_last":
        return base + coeff[2] * terms["quan"][-1] + tau[0] * terms["ent"][-1]
    if mode == "finetune":
        out = base + coeff[2] * np.sum(terms["quan"])
        for i, e in enumerate(terms["ent"]):
            out = out + tau[i] * e
        return out
    if mode == "one_ae_lpc":
        a, b = code_lens[0] / sum(code_lens), code_lens[1] / sum(code_lens)
        return base + coeff[2] * (lpc_terms["quan"] * a + terms["quan"][0] * b) + \
            tau[0] * (a * lpc_terms["ent"] + b * terms["ent"][0])
    if mode == "finetune_lpc":
        return base + coeff[2] * (lpc_terms["quan"] + np.sum(terms["quan"], axis=0))
    raise ValueError(mode)


def total_loss_sum(terms, coeff, tau, mode, **kw):
    """Scalar actually minimised: tf.gradients of a vector loss = gradient of its SUM [TF-semantics]."""
    return float(np.sum(phase_loss(terms, coeff, tau, mode, **kw)))


# --------------------------------------------------------------------------------------
# LPC utilities (lpc_utilities.py:28-77, 137-156) - audiolazy / spectrum restated, parity unpinned beyond the
# reference's own call sites (tests/golden/ref_env.py holds the same two third-party restatements)
# --------------------------------------------------------------------------------------
def lsf2poly(lsf):
    """spectrum.lsf2poly (Kondoz): roots e^{+-jw} alternate between the sum / difference polynomials; even order:
    P1 = P*(1 - z^-1), Q1 = Q*(1 + z^-1); a = (P1 + Q1)/2 without its last element."""
    lsf = np.asarray(lsf, np.float64)
    p = len(lsf)
    z = np.exp(1.0j * lsf)
    rQ, rP = z[0::2], z[1::2]
    Q = np.poly(np.concatenate((rQ, rQ.conjugate())))
    P = np.poly(np.concatenate((rP, rP.conjugate())))
    if p % 2:
        P1, Q1 = np.convolve(P, [1, 0, -1]), Q
    else:
        P1, Q1 = np.convolve(P, [1, -1]), np.convolve(Q, [1, 1])
    return np.real(0.5 * (P1 + Q1))[:-1]


def lsf2poly_after_quan(lpc_in_lsf, order=16):
    """lpc_utilities.py:28-33: per-frame lsf2poly, result cast to float32."""
    return np.stack([lsf2poly(r) for r in np.asarray(lpc_in_lsf)], 0).astype(np.float32)


def _fir_from_rest(a, x):
    """ZFilter(a)(x) from rest: e[n] = sum_k a[k] x[n-k], samples before the segment are zero."""
    return np.convolve(np.asarray(a, np.float64), np.asarray(x, np.float64))[:len(x)]


def lpc_analysis_get_residual(raw, poly):
    """lpc_utilities.py:37-77: seven 128-sample sub-frames at hop 64; each is filtered FROM REST by A(z) and
    cross-faded with hanning(128) (first: flat 64 then falling half; last: rising half then flat 64)."""
    raw = np.asarray(raw, np.float64).reshape(len(raw), FRAME_LENGTH)
    sub, half = FRAME_LENGTH // 4, FRAME_LENGTH // 8
    han = np.hanning(2 * half)
    out = np.zeros((len(raw), FRAME_LENGTH))
    for i in range(len(raw)):
        a = np.asarray(poly[i], np.float64)
        for j in range(7):
            if j == 0:
                w = np.append(np.ones(half), han[half:])
            elif j == 6:
                w = np.append(han[:half], np.ones(half))
            else:
                w = han
            s = j * half
            out[i, s:s + sub] += _fir_from_rest(a, raw[i, s:s + sub]) * w
    return out.astype(np.float32)


def lpc_synthesizer_tr(poly, res):
    """lpc_utilities.py:137-156: y = (1/A(z)) res per frame, from rest, double precision, cast to float32."""
    res = np.asarray(res, np.float64)
    out = np.zeros_like(res)
    for i in range(len(res)):
        a = np.asarray(poly[i], np.float64)
        y = np.zeros(res.shape[1])
        for n in range(res.shape[1]):
            acc = res[i, n]
            for k in range(1, min(len(a), n + 1)):
                acc -= a[k] * y[n - k]
            y[n] = acc / a[0]
        out[i] = y
    return out.astype(np.float32)


# --------------------------------------------------------------------------------------
# tau controllers (nsc_module:494-517 time domain; :630-639 LPC) - host logic restated for the value tests
# --------------------------------------------------------------------------------------
def tau_update(flag, tau, tau12, fully_entropy, ent_codec, target_entropy, lpc=False, is_quan_on=1.0):
    """Returns (tau, [tau_1, tau_2]) after one epoch.  ent_change = 0.015.
    time domain: finetune -> tau_i +-= 0.015 toward the hard-coded targets 1.5 / 2.5; else tau +-= 0.015 toward
    target_entropy.  LPC: only while is_quan_on == 1: +0.015 if H > target + 0.05, -0.045 if H < target."""
    ch = 0.015
    t1, t2 = tau12
    if lpc:
        if is_quan_on == 1.0:
            if fully_entropy > target_entropy + 0.05:
                tau += ch
            elif fully_entropy < target_entropy:
                tau -= ch * 3
        return tau, [t1, t2]
    if flag == "finetune":
        if ent_codec[0] > 1.5:
            t1 += ch
        if ent_codec[0] < 1.5:
            t1 -= ch
        if ent_codec[1] > 2.5:
            t2 += ch
        if ent_codec[1] < 2.5:
            t2 -= ch
    else:
        if fully_entropy > target_entropy:
            tau += ch
        elif fully_entropy < target_entropy:
            tau -= ch
    return tau, [t1, t2]
