"""CPU tests of the oracle itself: pinned against the reference-import KATs (tests/golden/reference_kats.json),
analytic known-answer tests for the TF-semantics restatements, and NumPy-f64 vs Torch-f64 agreement."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLD, "reference_kats.json")))
BKD = [9, 9, 100, 20, 1, 2]


# ---------------- pinned against the reference import ----------------
def test_framing_matches_reference():
    utt = np.random.default_rng(KATS["utt_seed"]).standard_normal(KATS["utt_len"])
    assert np.array_equal(O.utterance_to_segment(utt, True), np.array(KATS["seg_post"]))
    assert np.array_equal(O.utterance_to_segment(utt, False), np.array(KATS["seg_win"]))
    for n, cnt in KATS["frame_counts"].items():
        assert O.utterance_to_segment(np.zeros(int(n)), True).shape[0] == cnt


def test_hann_windows_match_reference():
    ones = np.ones(512)
    assert np.array_equal(O.hann_process(ones, 0, 3), np.array(KATS["hann_first"]))
    assert np.array_equal(O.hann_process(ones, 1, 3), np.array(KATS["hann_mid"]))
    assert np.array_equal(O.hann_process(ones, 2, 3), np.array(KATS["hann_last"]))
    assert abs(O.hann_process(ones, 1, 3).sum() - 480.0) < 1e-9


def test_scalar_helpers_match_reference():
    for e, s, v in KATS["entropy_to_bitrate"]:
        assert O.entropy_to_bitrate(e, s) == v
    for b, s, v in KATS["bitrate_to_entropy"]:
        assert O.bitrate_to_entropy(b, s) == v
    r0 = np.random.default_rng(0)
    a = r0.standard_normal(1000)
    b = a + 0.1 * r0.standard_normal(1000)
    assert O.snr(a, b)[1] == KATS["snr_seed0"]
    assert O.si_snr(b, a) == KATS["si_snr_seed0"]
    c = KATS["constants"]
    assert (O.INIT_ALPHA, O.FRAME_LENGTH, O.OVERLAP_EACH_SIDE, O.SAMPLE_RATE) == \
        (c["init_alpha"], c["frame_length"], c["overlap_each_side"], c["sample_rate"])


@pytest.mark.parametrize("key,strides", [("2", [2]), ("2_2", [2, 2])])
def test_topology_matches_reference_trace(key, strides):
    """The oracle's builders must create the same conv layers, in the same order, as the reference's
    _the_encoder_in_each_module/_the_decoder_in_each_module (captured with a recording tf stub)."""
    topo = KATS["topology"][key]
    ref_layers = []
    for op in topo["encoder"] + topo["decoder"]:
        if op[0] == "conv1d":
            ref_layers.append(("conv", op[3], op[1][2], op[2]))          # K, Cin, Cout
        elif op[0] == "separable_conv1d":
            ref_layers.append(("sep", op[3], op[1][2], op[2]))
    ps = O.ParamStore()
    out = O.codec_forward(np.zeros((1, 512, 1)), ps, "scope_1", BKD, strides, 32, 0.0, True)
    mine = []
    for name, v in ps.params.items():
        if name.endswith("/kernel"):
            mine.append(("conv",) + tuple(v.shape))
        elif name.endswith("/depthwise_kernel"):
            K, C, _ = v.shape
            mine.append(("sep", K, C, ps.params[name.replace("depthwise", "pointwise")].shape[2]))
    assert mine == ref_layers
    assert out["floating_code"].shape[1:] == tuple(topo["code_shape"][1:])
    n = sum(int(np.prod(v.shape)) for v in ps.params.values())
    assert n == {"2": 350185, "2_2": 540400}[key]  # SURVEY 3.2 / BASELINE.md


# ---------------- analytic KATs for the TF-semantics restatements ----------------
def test_same_pad_table():
    assert O.same_pad(512, 55) == (512, 27, 27)
    assert O.same_pad(512, 15, 2) == (512, 14, 14)
    assert O.same_pad(512, 15, 1) == (512, 7, 7)
    assert O.same_pad(512, 9) == (512, 4, 4)
    assert O.same_pad(512, 9, 1, 2) == (256, 3, 4)   # asymmetric for stride 2
    assert O.same_pad(511, 9, 1, 2) == (256, 4, 4)


def test_conv_delta_kernel_is_shift_and_stride2_probe():
    x = np.zeros((1, 512, 1)); x[0, 0, 0] = 1.0; x[0, 511, 0] = 2.0
    W = np.zeros((9, 1, 1)); W[0, 0, 0] = 1.0        # tap 0 reads x[t*2 - 3]
    y = O.conv1d(x, W, np.zeros(1), strides=2, activation=None)[0, :, 0]
    assert y.shape == (256,) and np.count_nonzero(y) == 0  # x[0] needs 2t-3=0 (no int t); x[511]: 2t-3=511 -> t=257 out
    W = np.zeros((9, 1, 1)); W[3, 0, 0] = 1.0        # tap 3 reads x[2t]
    y = O.conv1d(x, W, np.zeros(1), strides=2, activation=None)[0, :, 0]
    assert y[0] == 1.0 and np.count_nonzero(y) == 1
    W = np.zeros((9, 1, 1)); W[8, 0, 0] = 1.0        # tap 8 reads x[2t+5] -> 511 at t=253
    y = O.conv1d(x, W, np.zeros(1), strides=2, activation=None)[0, :, 0]
    assert y[253] == 2.0 and np.count_nonzero(y) == 1


def test_quantizer_limits():
    rng = np.random.default_rng(1)
    c = rng.uniform(-0.99, 0.99, (2, 16, 1))
    bins = np.linspace(-1, 1, 32)
    p, out = O.scalar_softmax_quantization(c, -300.0, bins, 0.0, True)
    assert np.array_equal(out, c)                     # is_quan_on = 0 -> identity
    p, out = O.scalar_softmax_quantization(c, -300.0, bins, 1.0, False)
    nearest = bins[np.argmin(np.abs(c - bins), axis=-1)]
    assert np.allclose(out[..., 0], nearest)
    p, out_soft = O.scalar_softmax_quantization(c, -3000.0, bins, 1.0, True)
    assert np.allclose(out_soft[..., 0], nearest, atol=1e-6)
    assert np.allclose(p.sum(-1), 1.0)
    # tie -> lowest index
    p, out = O.scalar_softmax_quantization(np.array([[[0.5]]]), -300.0, np.array([0.0, 1.0]), 1.0, False)
    assert out[0, 0, 0] == 0.0


def test_entropy_and_quan_loss_uniform():
    p = np.full((3, 8, 32), 1.0 / 32)
    assert abs(O.entropy_coding_loss(p) - 5.0) < 1e-4
    assert np.allclose(O.quan_loss(p), 32 * math.sqrt(1 / 32))


def test_rfft_single_bin_and_direct_dft():
    t = np.arange(512)
    st, mag = O.tf_stft(np.cos(2 * np.pi * 5 * t / 512)[None])
    assert abs(mag[0, 5] - 256.0) < 1e-6 and mag[0].argmax() == 5
    x = np.random.default_rng(3).standard_normal((3, 512))
    assert np.allclose(O.tf_stft(x)[0], O.rfft512_direct(x), atol=1e-9)


def test_mel_matrix_properties():
    for n in O.MEL_BANKS:
        M = O.linear_to_mel_weight_matrix(n)
        assert M.shape == (257, n) and np.all(M[0] == 0) and M.min() >= 0 and M.max() <= 1.0
    M128 = O.linear_to_mel_weight_matrix(128)
    assert (M128.sum(0) == 0).sum() > 0           # 128-bank has empty low filters
    M8 = O.linear_to_mel_weight_matrix(8)
    assert np.all(M8.sum(0) > 0)


def test_adam_tf1_first_step():
    g = np.array([0.5, -2.0, 0.0])
    th, m, v = O.adam_tf1_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1, 1e-3)
    expect = -1e-3 * g / (np.abs(g) + 1e-8 / math.sqrt(0.001))
    assert np.allclose(th, expect, rtol=1e-9, atol=0) and th[2] == 0.0


# ---------------- NumPy-f64 oracle == Torch-f64 oracle ----------------
def _setup(strides, B=2, nb=32, seed=1234):
    ps = O.ParamStore(np.random.default_rng(20200504))
    rng = np.random.default_rng(seed)
    x = np.clip(0.03 * rng.standard_normal((B, 512, 1)), -1, 1) * O.training_window()[None, :, None]
    # random biases / non-degenerate alpha so every term is exercised
    out = O.codec_forward(x, ps, "scope_1", BKD, strides, nb, 1.0, True)
    for k in ps.params:
        if k.endswith("/bias"):
            ps.params[k] = (0.05 * rng.standard_normal(ps.params[k].shape)).astype(np.float32).astype(np.float64)
    ps.params["scope_1/alpha"] = np.array(-20.0)
    return ps, x


@pytest.mark.parametrize("strides", [[2], [2, 2]])
def test_numpy_and_torch_oracles_agree_forward(strides):
    ps, x = _setup(strides)
    ps.begin_replay()
    o = O.codec_forward(x, ps, "scope_1", BKD, strides, 32, 1.0, True)
    tp = OT.TorchParams(ps, requires_grad=False)
    ot = OT.codec_forward(torch.tensor(x), tp, "scope_1", BKD, strides, 1.0, True)
    for k in ("p", "floating_code", "code", "decoded"):
        a, b = o[k], ot[k].numpy()
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a))), k
    tgt = x[:, :, 0]
    assert np.allclose(O.mse_loss(o["decoded"], tgt), OT.mse_loss(ot["decoded"], torch.tensor(tgt)).numpy(), rtol=1e-10)
    assert np.allclose(O.mfcc_loss(o["decoded"], tgt), OT.mfcc_loss(ot["decoded"], torch.tensor(tgt)).numpy(), rtol=1e-9)
    assert np.allclose(O.quan_loss(o["p"]), OT.quan_loss(ot["p"]).numpy(), rtol=1e-10)
    assert abs(O.entropy_coding_loss(o["p"]) - float(OT.entropy_coding_loss(ot["p"]))) < 1e-10
    # hard path
    ps.begin_replay()
    oh = O.codec_forward(x, ps, "scope_1", BKD, strides, 32, 1.0, False)
    tp.reset()
    oth = OT.codec_forward(torch.tensor(x), tp, "scope_1", BKD, strides, 1.0, False)
    assert np.max(np.abs(oh["decoded"] - oth["decoded"].numpy())) < 1e-9


def test_torch_gradients_match_numpy_finite_differences():
    """Pins the autograd oracle's gradients to the NumPy oracle's forward (directional derivatives)."""
    strides = [2]
    ps, x = _setup(strides, B=2)
    coeff, tau = [60.0, 10.0, 10.0, 0.0], 0.3
    tgt = x[:, :, 0]

    def f_numpy():
        ps.begin_replay()
        o = O.codec_forward(x, ps, "scope_1", BKD, strides, 32, 1.0, True)
        terms = O.loss_terms(o["decoded"], tgt, [o["p"]])
        return O.total_loss_sum(terms, coeff, tau, "quan_last")

    tp = OT.TorchParams(ps)
    ot = OT.codec_forward(torch.tensor(x), tp, "scope_1", BKD, strides, 1.0, True)
    loss = OT.total_loss_sum(ot["decoded"], torch.tensor(tgt), [ot["p"]], coeff, tau, "quan_last")
    assert abs(float(loss) - f_numpy()) < 1e-8 * abs(float(loss))
    loss.backward()
    rng = np.random.default_rng(5)
    names = ["scope_1/alpha", "scope_1/bins", "scope_1/conv1d/kernel", "scope_1/conv1d_4/kernel",
             "scope_1/conv1d_9/kernel", "scope_1/conv1d_17/bias", "scope_1/conv1d_18/kernel",
             "scope_1/separable_conv1d/depthwise_kernel", "scope_1/separable_conv1d/pointwise_kernel",
             "scope_1/conv1d_35/kernel"]
    for n in names:
        g = tp.t[n].grad.numpy()
        d = rng.standard_normal(g.shape)
        d /= np.linalg.norm(d) + 1e-30
        base = ps.params[n].copy()
        eps = 1e-5 * max(1.0, float(np.max(np.abs(base))))
        ps.params[n] = base + eps * d
        fp = f_numpy()
        ps.params[n] = base - eps * d
        fm = f_numpy()
        ps.params[n] = base
        fd = (fp - fm) / (2 * eps)
        an = float(np.sum(g * d))
        assert abs(fd - an) <= 2e-5 * max(abs(an), abs(fd), 1e-3), (n, fd, an)


def test_committed_codec_golden_matches_oracle():
    gold = np.load(os.path.join(GOLD, "codec_golden.npz"))
    ps, x = _setup([2], B=int(gold["x"].shape[0]))
    assert np.array_equal(x, gold["x"])
    ps.begin_replay()
    o = O.codec_forward(x, ps, "scope_1", BKD, [2], 32, 1.0, True)
    assert np.allclose(o["decoded"], gold["decoded"], rtol=0, atol=1e-12)
    assert np.allclose(o["floating_code"], gold["floating_code"], rtol=0, atol=1e-12)
    assert np.allclose(O.mfcc_loss(o["decoded"], x[:, :, 0]), gold["freq_loss"], rtol=1e-12)
