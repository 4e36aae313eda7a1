"""CPU ORACLE (float64 NumPy) for the NSC/CMRL hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path (``nsc_amd``)
must never import anything from ``oracle/``.

PARITY STATUS: **parity unpinned** for every TensorFlow-executed op.  The reference
(cocosci/NSC) ships no tests, golden vectors or checkpoints, and TensorFlow is not
installable in the build container, so the TF op semantics below ([TF-semantics] tags)
are restated from TF's documented behaviour.  What *is* pinned against the reference
itself (by importing its pure-NumPy helpers, see tests/golden/make_reference_fixtures.py):
frame indexing / Hann windows (utilities.py), entropy<->bitrate helpers, snr / si_snr,
the LSF bin table, and the layer topology trace of the encoder/decoder builders.

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
class ParamStore:
    """Ordered name->array store with TF1-style auto-uniquified layer names
    (conv1d, conv1d_1, ...; separable_conv1d, ...) inside a scope (nsc_module:267)."""

    def __init__(self, rng=None):
        self.params = OrderedDict()
        self._counts = {}
        self.rng = rng if rng is not None else np.random.default_rng(20200504)
        self.replay = False  # when True, get() returns existing params in creation order

    def _uniq(self, scope, base):
        key = (scope, base)
        n = self._counts.get(key, 0)
        self._counts[key] = n + 1
        return f"{scope}/{base}" if n == 0 else f"{scope}/{base}_{n}"

    def begin_replay(self):
        self._counts = {}
        self.replay = True

    def glorot(self, shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        # float32-representable values so every implementation starts from identical weights
        return self.rng.uniform(-lim, lim, size=shape).astype(np.float32).astype(np.float64)

    def conv(self, scope, K, Cin, Cout):
        name = self._uniq(scope, "conv1d")
        if not self.replay:
            # glorot_uniform fans for a [K,Cin,Cout] kernel: K*Cin / K*Cout [TF-semantics]
            self.params[name + "/kernel"] = self.glorot((K, Cin, Cout), K * Cin, K * Cout)
            self.params[name + "/bias"] = np.zeros(Cout)
        return self.params[name + "/kernel"], self.params[name + "/bias"]

    def sepconv(self, scope, K, C, Cout):
        name = self._uniq(scope, "separable_conv1d")
        if not self.replay:
            # Keras: depthwise [K,C,1] -> fan_in=K*C, fan_out=K*1; pointwise [1,C,Cout] -> C / Cout
            self.params[name + "/depthwise_kernel"] = self.glorot((K, C, 1), K * C, K)
            self.params[name + "/pointwise_kernel"] = self.glorot((1, C, Cout), C, Cout)
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
    bins = ps.var(scope, "bins", np.linspace(-BETA_BOUNDARY, BETA_BOUNDARY, num_bins))
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


def hertz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)


def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins=257, sample_rate=16000,
                                lower_edge_hertz=0.0, upper_edge_hertz=8000.0):
    """tf.signal.linear_to_mel_weight_matrix [TF-semantics] (loss_terms_and_measures.py:138):
    HTK mel scale, DC bin zeroed, triangles built in the mel domain, computed in float64."""
    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]
    spec_mel = hertz_to_mel(lin)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz), hertz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[1, 0], [0, 0]])


MEL_BANKS = (8, 16, 32, 128)  # loss_terms_and_measures.py:133


def mel_matrix_cat():
    """The 4 banks concatenated to [257, 184]; cast through float32 like TF's returned matrix."""
    return np.concatenate([linear_to_mel_weight_matrix(n) for n in MEL_BANKS], axis=1) \
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


def total_loss_sum(terms, coeff, tau, mode):
    """Scalar actually minimised: losses are [B] vectors, tf.gradients of a vector = gradient of its SUM
    [TF-semantics] (nsc_module:914-926; cmrl.py:101-113, 355-372).
    mode: 'no_quan' | 'quan_last' (one_ae / followers: newest codec only) | 'finetune' (sum quan, tau_i*ent_i)
          | 'finetune_lpc' (sum of quan terms, no entropy term: cmrl.py:483-485)."""
    B = terms["time"].shape[0]
    tot = np.sum(coeff[0] * terms["time"] + coeff[1] * terms["freq"])
    if mode == "no_quan":
        return tot
    if mode == "quan_last":
        return tot + np.sum(coeff[2] * terms["quan"][-1]) + B * float(np.ravel(tau)[0]) * terms["ent"][-1]
    if mode == "finetune":
        tau = np.ravel(tau)
        tot += np.sum(coeff[2] * np.sum(terms["quan"], axis=0))
        for i, e in enumerate(terms["ent"]):
            tot += B * tau[i] * e
        return tot
    if mode == "finetune_lpc":
        return tot + np.sum(coeff[2] * np.sum(terms["quan"], axis=0))
    raise ValueError(mode)
