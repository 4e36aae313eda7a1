"""CPU ORACLE #2 (PyTorch-CPU, autograd) for the NSC/CMRL hot path.  TEST INFRASTRUCTURE ONLY.

Independent second restatement of the same graph as ``oracle/nsc_oracle.py`` (float64 NumPy).
It exists for two things only:
  * gradients (autograd, float64) and TF1-Adam reference steps for the parity tests;
  * the ``cpu_baseline`` leg of ``bench.py`` (float32, "CPU restatement of the reference TF graph").
The product path (``nsc_amd``) must never import it.  PARITY STATUS: parity unpinned for the
TF-executed ops (see nsc_oracle.py header); the two oracles are pinned to each other
(tests/test_oracle.py, <= 1e-9 rel in float64).

Tensors are channels_last ``[B,T,C]`` at the function boundaries, like the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nsc_oracle as O


LRELU_KINK_SHIFT = 0.0     # see act(): 0 = the reference's leaky_relu


def _pad_same(x_bct, K, dil, stride):
    T = x_bct.shape[-1]
    t_out, pl, pr = O.same_pad(T, K, dil, stride)
    return F.pad(x_bct, (pl, pr)), t_out


def act(x, activation):
    if activation is None or activation == "none":
        return x
    if activation == "tanh":
        return torch.tanh(x)
    if activation == "lrelu":
        if LRELU_KINK_SHIFT:
            # test instrument (tests/test_reference_exec_gpu.py::_kink_sensitivity): the kink moved by a fraction of the tensor's rms,
            # so that every pre-activation within that distance of zero takes the OTHER slope
            thr = LRELU_KINK_SHIFT * x.detach().pow(2).mean().sqrt()
            return torch.where(x > thr, x, O.LRELU_ALPHA * x)
        return F.leaky_relu(x, O.LRELU_ALPHA)
    raise ValueError(activation)


def conv1d(x, W, b, dilation_rate=1, strides=1, activation="tanh"):
    """nn_core_operator.py:6-14; W is TF layout [K,Cin,Cout]."""
    xb = x.transpose(1, 2)
    xp, _ = _pad_same(xb, W.shape[0], dilation_rate, strides)
    y = F.conv1d(xp, W.permute(2, 1, 0), b, stride=strides, dilation=dilation_rate)
    return act(y.transpose(1, 2), activation)


def conv1d_depth(x, Wd, Wp, b, activation=None):
    """nn_core_operator.py:17-21 (SeparableConv1D)."""
    C = x.shape[-1]
    xb = x.transpose(1, 2)
    xp, _ = _pad_same(xb, Wd.shape[0], 1, 1)
    dw = F.conv1d(xp, Wd.permute(1, 2, 0), None, groups=C)  # [C,1,K]
    y = F.conv1d(dw, Wp.permute(2, 1, 0), b)
    return act(y.transpose(1, 2), activation)


def subpixel_shuffle(x, s=2):
    B, T, C = x.shape
    return x.reshape(B, T, C // s, s).permute(0, 1, 3, 2).reshape(B, T * s, C // s)


class TorchParams:
    """Torch view of an oracle ParamStore, consumed in creation order (like TF scopes)."""

    def __init__(self, ps: O.ParamStore, dtype=torch.float64, requires_grad=True):
        self.names = list(ps.params.keys())
        self.t = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad)
                  for k, v in ps.params.items()}
        self._counts = {}

    def reset(self):
        self._counts = {}

    def _uniq(self, scope, base):
        key = (scope, base)
        n = self._counts.get(key, 0)
        self._counts[key] = n + 1
        return f"{scope}/{base}" if n == 0 else f"{scope}/{base}_{n}"

    def conv(self, scope):
        n = self._uniq(scope, "conv1d")
        return self.t[n + "/kernel"], self.t[n + "/bias"]

    def sepconv(self, scope):
        n = self._uniq(scope, "separable_conv1d")
        return self.t[n + "/depthwise_kernel"], self.t[n + "/pointwise_kernel"], self.t[n + "/bias"]

    def scope_tensors(self, scope):
        return [self.t[k] for k in self.names if k.startswith(scope + "/")]


def gated_bottleneck(x, tp, scope, dilation_rate, is_last_flat):
    W1, b1 = tp.conv(scope)
    h = act(conv1d(x, W1, b1, activation=None), "lrelu")
    Wl, bl = tp.conv(scope)
    left = conv1d(h, Wl, bl, dilation_rate=dilation_rate, activation=None)
    Wr, br = tp.conv(scope)
    right = conv1d(h, Wr, br, dilation_rate=dilation_rate, activation="tanh")
    W9, b9 = tp.conv(scope)
    y = conv1d(left * right, W9, b9, activation=None) + x
    return y if is_last_flat else act(y, "lrelu")


def stack(x, tp, scope, bkd):
    n = len(bkd) - 4
    for i in range(n):
        x = gated_bottleneck(x, tp, scope, bkd[i + 4], i == n - 1)
    return x


def encoder(x, tp, scope, bkd, strides):
    W, b = tp.conv(scope)
    h = act(conv1d(x, W, b, activation=None), "lrelu")
    for s in strides:
        h = stack(h, tp, scope, bkd)
        W, b = tp.conv(scope)
        h = act(conv1d(h, W, b, strides=s, activation=None), "lrelu")
    h = stack(h, tp, scope, bkd)
    W, b = tp.conv(scope)
    return conv1d(h, W, b, activation="tanh")


def decoder(code, tp, scope, bkd, strides):
    h = code
    for s in strides:
        h = stack(h, tp, scope, bkd)
        Wd, Wp, b = tp.sepconv(scope)
        h = subpixel_shuffle(act(conv1d_depth(h, Wd, Wp, b), "lrelu"), s)
    h = stack(h, tp, scope, bkd)
    W, b = tp.conv(scope)
    return conv1d(h, W, b, activation=None)


def scalar_softmax_quantization(code, alpha, bins, is_quan_on, the_share):
    dist = (code - bins.reshape(1, 1, -1)).abs()
    p = torch.softmax(alpha * dist, dim=-1)
    if the_share:
        sel = p
    else:
        # lowest index among ties (tf.nn.top_k): argmax on the reversed axis picks the last max there
        nb = bins.numel()
        idx = (nb - 1) - torch.argmax(torch.flip(p, dims=[-1]), dim=-1)
        sel = F.one_hot(idx, nb).to(p.dtype)
    q = (sel @ bins).unsqueeze(-1)
    return p, (1.0 - is_quan_on) * code + is_quan_on * q


def codec_forward(x, tp, scope, bkd, strides, is_quan_on, the_share):
    alpha, bins = tp.t[scope + "/alpha"], tp.t[scope + "/bins"]
    code = encoder(x, tp, scope, bkd, strides)
    p, q = scalar_softmax_quantization(code, alpha, bins, is_quan_on, the_share)
    dec = decoder(q, tp, scope, bkd, strides)
    return dict(p=p, floating_code=code, code=q, decoded=dec[:, :, 0])


def mse_loss(d, o):
    return torch.sqrt(torch.mean((d - o) ** 2, dim=-1) + 1e-7)


_MEL_CACHE = {}


def _mel(dtype):
    if dtype not in _MEL_CACHE:
        _MEL_CACHE[dtype] = torch.tensor(O.mel_matrix_cat(), dtype=dtype)
    return _MEL_CACHE[dtype]


def mfcc_loss(d, o):
    def logmel(sig):
        st = torch.fft.rfft(sig.reshape(-1, O.FRAME_LENGTH), n=O.FRAME_LENGTH, dim=-1)
        mag = torch.sqrt(st.real ** 2 + st.imag ** 2 + 1e-7)
        psd = mag ** 2 / O.FRAME_LENGTH
        return torch.log(psd @ _mel(sig.dtype) + 1e-7)

    dl, ol = logmel(d), logmel(o)
    outs, off = [], 0
    for n in O.MEL_BANKS:
        outs.append(torch.sqrt(torch.mean((dl[:, off:off + n] - ol[:, off:off + n]) ** 2, dim=-1) + 1e-7))
        off += n
    return torch.stack(outs, -1).mean(-1)


def quan_loss(p):
    return torch.sqrt(p + 1e-20).sum(-1).mean(-1)


def entropy_coding_loss(p, hist_extra=None):
    h = p.reshape(-1, p.shape[-1]).sum(0)
    if hist_extra is not None:
        h = h + hist_extra
    h = h / h.sum()
    return -(h * torch.log(h + 1e-7) / math.log(2.0)).sum()


def cascade_forward(x, tp, bkd, strides_per_codec, is_quan_on, the_share, res_scalar=1.0, scale_first=False,
                    frozen=()):
    """cmrl.py cascade wiring; ``frozen`` codec indices run under no_grad-like detach of their params
    (followers: earlier scopes are forward-only, cmrl.py:106-113)."""
    tp.reset()
    outs, yhat = [], []
    for i, st in enumerate(strides_per_codec):
        if i == 0 and not scale_first:
            o = codec_forward(x, tp, f"scope_{i + 1}", bkd, st, is_quan_on, the_share)
            y = o["decoded"]
        else:
            xin = x if i == 0 else x - torch.stack(yhat, 0).sum(0).unsqueeze(-1)
            o = codec_forward(res_scalar * xin, tp, f"scope_{i + 1}", bkd, st, is_quan_on, the_share)
            y = o["decoded"] / res_scalar
        yhat.append(y)
        outs.append(o)
    return outs, torch.stack(yhat, 0).sum(0)


def phase_loss(decoded, target, p_list, coeff, tau, mode, p_extra=(), code_lens=(16.0, 256.0)):
    """The [B] loss vector a phase minimises (see nsc_oracle.phase_loss for the reference lines).
    p_extra: (p_lpc,) - the LSF quantizer's soft assignment for the two LPC modes."""
    base = coeff[0] * mse_loss(decoded, target) + coeff[1] * mfcc_loss(decoded, target)
    tau = np.ravel(np.asarray(tau, dtype=np.float64))
    if mode == "no_quan":
        return base
    if mode == "quan_last":
        return base + coeff[2] * quan_loss(p_list[-1]) + float(tau[0]) * entropy_coding_loss(p_list[-1])
    if mode == "finetune":
        out = base + coeff[2] * torch.stack([quan_loss(p) for p in p_list], 0).sum()   # scalar (cmrl.py:355)
        for i, p in enumerate(p_list):
            out = out + float(tau[i]) * entropy_coding_loss(p)
        return out
    if mode == "one_ae_lpc":
        a, b = code_lens[0] / sum(code_lens), code_lens[1] / sum(code_lens)
        pl = p_extra[0]
        return base + coeff[2] * (quan_loss(pl) * a + quan_loss(p_list[0]) * b) + \
            float(tau[0]) * (a * entropy_coding_loss(pl) + b * entropy_coding_loss(p_list[0]))
    if mode == "finetune_lpc":
        out = base
        for p in list(p_extra) + list(p_list):
            out = out + coeff[2] * quan_loss(p)
        return out
    raise ValueError(mode)


def total_loss_sum(decoded, target, p_list, coeff, tau, mode, p_extra=(), code_lens=(16.0, 256.0)):
    """Scalar minimised by the reference's optimizers: the SUM of the [B] loss vector [TF-semantics]."""
    return phase_loss(decoded, target, p_list, coeff, tau, mode, p_extra, code_lens).sum()


def adam_tf1_step_(params, grads, ms, vs, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place TF1 Adam over lists of tensors (see nsc_oracle.adam_tf1_step)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, ms, vs):
            m.mul_(beta1).add_(g, alpha=1.0 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            p.sub_(lr_t * m / (v.sqrt() + eps))
