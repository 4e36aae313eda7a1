"""GPU parity tests of every C-ABI kernel against the CPU oracle (called through ctypes on raw device pointers)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT
from tests._util import assert_close, dev, relerr

pytestmark = pytest.mark.gpu


class _PoisonedLib:
    """The C-ABI library with every launch preceded by _poison_lds_with_nan: each kernel under test starts on LDS full of NaN
    instead of on whatever the previous kernel left there, so a read of LDS the kernel never wrote (a table pad, a row or
    channel group past the tensor) shows up as NaN x 0 = NaN in its output - every time, not when the stale bytes fall badly."""
    _PLAIN = ("nsc_last_error", "nsc_version", "nsc_gated_block_image_floats", "nsc_gated_block_image_index", "nsc_gated_block_pair_flag_ints",
              "nsc_conv1d_wgrad_workspace", "nsc_gated_block_wgrad_batch_workspace", "nsc_conv1d_wgrad_batch_workspace",
              "nsc_gated_block_wgrad_workspace")

    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in self._PLAIN or not name.startswith("nsc_"):
            return fn
        real = self._real

        def call(*a, **k):
            _poison_lds_with_nan(real)
            return fn(*a, **k)
        return call

    def __setattr__(self, name, value):
        setattr(self._real, name, value)


@pytest.fixture(scope="module")
def lib():
    from nsc_amd import _lib
    return _PoisonedLib(_lib.load())


def _desc(**kw):
    from nsc_amd._lib import ConvDesc
    base = dict(B=1, Cin=1, Cout=1, Tin=1, Tout=1, K=1, dil=1, stride=1, padL=0, act=0, res_mode=0, mul_mode=0,
                out_mode=0, in_up=0, accumulate=0)
    base.update(kw)
    return ConvDesc(**base)


def _st():
    return torch.cuda.current_stream().cuda_stream


_KEEP = []


def P(a):
    """Upload a host array and return its device pointer; the tensor is kept alive until the test module ends
    (a temporary's memory could be recycled by the caching allocator before the asynchronous kernel runs)."""
    t = a if isinstance(a, torch.Tensor) else dev(a)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t.data_ptr()


ACT = {"none": 0, "tanh": 1, "lrelu": 2}

CONV_CASES = [
    # B, Cin, Cout, T, K, dil, stride, act, res_mode
    (3, 1, 100, 512, 55, 1, 1, "lrelu", 0),
    (2, 100, 20, 512, 1, 1, 1, "lrelu", 0),
    (2, 20, 20, 512, 15, 2, 1, "none", 0),
    (2, 20, 20, 256, 15, 1, 1, "tanh", 0),
    (2, 20, 100, 512, 9, 1, 1, "lrelu", 1),
    (2, 20, 100, 256, 9, 1, 1, "none", 2),
    (2, 100, 100, 512, 9, 1, 2, "lrelu", 0),
    (2, 50, 50, 512, 9, 1, 1, "none", 1),
    (2, 1, 20, 256, 1, 1, 1, "lrelu", 0),
    (1, 100, 1, 256, 55, 1, 1, "tanh", 0),
    (2, 50, 1, 512, 55, 1, 1, "none", 0),
    (130, 12, 1, 500, 55, 1, 1, "tanh", 1),   # Cout = 1, enough tiles for the 4-outputs-per-lane register-tiled kernel
    (3, 9, 1, 100, 55, 1, 1, "none", 0),      # Cout = 1, T < one tile, channels not a multiple of the 8 waves
    (2, 7, 13, 300, 5, 3, 1, "tanh", 0),      # ragged: odd channels, T not a tile multiple
    (2, 6, 130, 77, 3, 1, 2, "none", 0),      # Cout > 112 (grid.z), odd T with stride 2
    (1, 100, 100, 128, 1, 1, 1, "lrelu", 0),  # pointwise
    (2, 100, 200, 200, 5, 1, 1, "none", 1),   # 32x32x2 kernel (Cout >= 96): two 112-row groups, ragged second time tile
    (2, 6, 130, 300, 3, 2, 2, "tanh", 2),     # 32x32x2 kernel: 112 + 18 rows, stride 2 with dilation, T_out = 150
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_fwd(lib, case):
    B, Cin, Cout, T, K, dil, s, act, res_mode = case
    rng = np.random.default_rng(sum(int(v) * (i + 3) for i, v in enumerate(case) if not isinstance(v, str)))
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    W = (rng.standard_normal((K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    Tout, padL, _ = O.same_pad(T, K, dil, s)
    ref = O.conv1d(x, W, b, dil, s, None)
    res = None
    if res_mode == 1:
        res = rng.standard_normal((B, Tout, Cout)).astype(np.float32)
        ref = ref + res
    elif res_mode == 2:
        res = rng.standard_normal((B, Tout, 1)).astype(np.float32)
        ref = ref + res
    ref = O._act(ref, act)
    xd, wd, bd = dev(x.transpose(0, 2, 1)), dev(W), dev(b)
    rd = dev(res.transpose(0, 2, 1)) if res is not None else None
    y = torch.full((B, Cout, Tout), float("nan"), device="cuda")
    d = _desc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=Tout, K=K, dil=dil, stride=s, padL=padL, act=ACT[act], res_mode=res_mode)
    fn = lib.nsc_conv1d_cout1_fwd if Cout == 1 else lib.nsc_conv1d_fwd
    rc = fn(C.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None, None,
            y.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy().transpose(0, 2, 1), ref, what=f"conv fwd {case}")


def test_conv1d_fwd_generic_kernel_with_cout1(lib):
    """The MFMA kernel itself must also be right for Cout == 1 (it is the dgrad path of Cin == 1 convs)."""
    rng = np.random.default_rng(5)
    B, Cin, T, K = 2, 20, 256, 15
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    W = rng.standard_normal((K, Cin, 1)).astype(np.float32)
    ref = O.conv1d(x, W, None, 2, 1, None)
    y = torch.empty((B, 1, T), device="cuda")
    d = _desc(B=B, Cin=Cin, Cout=1, Tin=T, Tout=T, K=K, dil=2, padL=14)
    assert lib.nsc_conv1d_fwd(C.byref(d), P(x.transpose(0, 2, 1)), P(W), None, None, None,
                              y.data_ptr(), _st()) == 0
    assert_close(y.cpu().numpy().transpose(0, 2, 1), ref, what="generic cout1")


def test_conv1d_shuffle_and_mul_epilogues(lib):
    rng = np.random.default_rng(9)
    B, C_, T = 2, 100, 256
    x = rng.standard_normal((B, T, C_)).astype(np.float32)
    W = (rng.standard_normal((1, C_, C_)) / 10).astype(np.float32)
    b = rng.standard_normal(C_).astype(np.float32)
    ref = O.subpixel_shuffle(O.conv1d(x, W, b, 1, 1, "lrelu"), 2)          # [B, 512, 50]
    y = torch.empty((B, C_ // 2, 2 * T), device="cuda")
    d = _desc(B=B, Cin=C_, Cout=C_, Tin=T, Tout=T, K=1, act=2, out_mode=1)
    xd = dev(x.transpose(0, 2, 1))
    assert lib.nsc_conv1d_fwd(C.byref(d), xd.data_ptr(), P(W), P(b), None, None,
                              y.data_ptr(), _st()) == 0
    assert_close(y.cpu().numpy().transpose(0, 2, 1), ref, what="shuffle epilogue")
    # unshuffle is the inverse permutation
    back = torch.empty((B, C_, T), device="cuda")
    assert lib.nsc_unshuffle2(y.data_ptr(), back.data_ptr(), B, C_, T, _st()) == 0
    assert_close(back.cpu().numpy().transpose(0, 2, 1), O.conv1d(x, W, b, 1, 1, "lrelu"), what="unshuffle")
    # mul_mode: v * lrelu'(aux) and v * tanh'(aux)
    aux = rng.standard_normal((B, T, C_)).astype(np.float32)
    for mode, g in ((1, np.where(aux > 0, 1.0, 0.2)), (2, 1 - aux.astype(np.float64) ** 2)):
        d = _desc(B=B, Cin=C_, Cout=C_, Tin=T, Tout=T, K=1, mul_mode=mode)
        y2 = torch.empty((B, C_, T), device="cuda")
        assert lib.nsc_conv1d_fwd(C.byref(d), xd.data_ptr(), P(W), P(b), None,
                                  P(aux.transpose(0, 2, 1)), y2.data_ptr(), _st()) == 0
        assert_close(y2.cpu().numpy().transpose(0, 2, 1), O.conv1d(x, W, b, 1, 1, None) * g, what=f"mul_mode {mode}")


GRAD_CASES = [
    (3, 1, 100, 512, 55, 1, 1), (2, 100, 20, 512, 1, 1, 1), (2, 20, 20, 512, 15, 2, 1), (2, 20, 100, 256, 9, 1, 1),
    (2, 100, 100, 512, 9, 1, 2), (2, 100, 1, 256, 55, 1, 1), (2, 50, 1, 512, 55, 1, 1), (2, 1, 20, 256, 1, 1, 1),
    (2, 7, 13, 300, 5, 3, 1), (3, 50, 50, 512, 15, 1, 1), (2, 100, 100, 256, 1, 1, 1), (40, 100, 100, 512, 9, 1, 2),
    (1, 3, 5, 40, 3, 1, 1),
]


@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv1d_dgrad_wgrad(lib, case):
    """dgrad = forward kernel on nsc_weight_flip_transpose'd weights; wgrad incl. bias row; vs autograd oracle."""
    B, Cin, Cout, T, K, dil, s = case
    rng = np.random.default_rng(sum(int(v) * (i + 3) for i, v in enumerate(case) if not isinstance(v, str)))
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    W = (rng.standard_normal((K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    b = np.zeros(Cout, np.float32)
    Tout, padL, _ = O.same_pad(T, K, dil, s)
    dz = rng.standard_normal((B, Tout, Cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    (OT.conv1d(xt, Wt, bt, dil, s, None) * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    xd, wd, dzd = dev(x.transpose(0, 2, 1)), dev(W), dev(dz.transpose(0, 2, 1))
    # ---- wgrad ----
    dw = torch.zeros((K, Cin, Cout), device="cuda")
    db = torch.zeros((Cout,), device="cuda")
    if Cout == 1:   # role swap (see nsc_amd/engine.py _Conv.wgrad)
        d = _desc(B=B, Cin=1, Cout=Cin, Tin=Tout, Tout=T, K=K, dil=dil, padL=(K - 1) * dil - padL)
        assert lib.nsc_conv1d_wgrad(C.byref(d), dzd.data_ptr(), xd.data_ptr(), dw.data_ptr(), None, 1, _st()) == 0, lib.nsc_last_error()
        assert lib.nsc_sum_all(dzd.data_ptr(), db.data_ptr(), dzd.numel(), _st()) == 0
    else:
        d = _desc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=Tout, K=K, dil=dil, stride=s, padL=padL)
        assert lib.nsc_conv1d_wgrad(C.byref(d), xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, _st()) == 0, lib.nsc_last_error()
    assert_close(dw.cpu().numpy(), Wt.grad.numpy(), what=f"wgrad {case}")
    assert_close(db.cpu().numpy(), bt.grad.numpy(), what=f"bias grad {case}")
    # ---- wgrad again through the slab (store + reduce) flush: poisoned workspace, accumulates on top of dw ----
    lib.nsc_conv1d_wgrad_workspace.restype = C.c_long
    nws = lib.nsc_conv1d_wgrad_workspace(C.byref(d))
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device="cuda")
    dw2, db2 = dw.clone(), db.clone()
    if Cout == 1:
        assert lib.nsc_conv1d_wgrad_ws(C.byref(d), dzd.data_ptr(), xd.data_ptr(), dw2.data_ptr(), None, 1, ws.data_ptr(), nws, _st()) == 0
    else:
        assert lib.nsc_conv1d_wgrad_ws(C.byref(d), xd.data_ptr(), dzd.data_ptr(), dw2.data_ptr(), db2.data_ptr(), 0, ws.data_ptr(), nws, _st()) == 0
        assert_close(db2.cpu().numpy(), 2 * bt.grad.numpy(), what=f"bias grad (slab) {case}")
    assert_close(dw2.cpu().numpy(), 2 * Wt.grad.numpy(), what=f"wgrad (slab) {case}")
    # ---- dgrad ----
    wt = torch.empty((K, Cout, Cin), device="cuda")
    assert lib.nsc_weight_flip_transpose(wd.data_ptr(), wt.data_ptr(), K, Cin, Cout, _st()) == 0
    assert np.array_equal(wt.cpu().numpy(), W[::-1].transpose(0, 2, 1))
    dx = torch.full((B, Cin, T), float("nan"), device="cuda")
    d = _desc(B=B, Cin=Cout, Cout=Cin, Tin=Tout, Tout=T, K=K, dil=dil, stride=1, padL=(K - 1) * dil - padL,
              in_up=1 if s == 2 else 0)
    fn = lib.nsc_conv1d_cout1_fwd if Cin == 1 else lib.nsc_conv1d_fwd
    assert fn(C.byref(d), dzd.data_ptr(), wt.data_ptr(), None, None, None, dx.data_ptr(), _st()) == 0, lib.nsc_last_error()
    assert_close(dx.cpu().numpy().transpose(0, 2, 1), xt.grad.numpy(), what=f"dgrad {case}")


@pytest.mark.parametrize("shape", [(3, 100, 256, 9), (9, 50, 128, 9), (6, 20, 520, 5), (2, 12, 1024, 9)])
def test_depthwise_fwd_bwd(lib, shape):
    """(the weight gradient takes the four-frames-in-flight kernel up to T = 752, the frame-at-a-time one above that)"""
    rng = np.random.default_rng(11)
    B, C_, T, K = shape
    x = rng.standard_normal((B, T, C_)).astype(np.float32)
    Wd = rng.standard_normal((K, C_, 1)).astype(np.float32)
    dy = rng.standard_normal((B, T, C_)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(Wd, dtype=torch.float64, requires_grad=True)
    eye = torch.eye(C_, dtype=torch.float64).reshape(1, C_, C_)
    yt = OT.conv1d_depth(xt, wt, eye, torch.zeros(C_, dtype=torch.float64))
    (yt * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    xd, wdd, dyd = dev(x.transpose(0, 2, 1)), dev(Wd.reshape(K, C_)), dev(dy.transpose(0, 2, 1))
    y = torch.empty((B, C_, T), device="cuda")
    assert lib.nsc_depthwise_fwd(xd.data_ptr(), wdd.data_ptr(), y.data_ptr(), B, C_, T, K, _st()) == 0
    assert_close(y.cpu().numpy().transpose(0, 2, 1), yt.detach().numpy(), what="depthwise fwd")
    dx = torch.empty((B, C_, T), device="cuda")
    dwd = torch.zeros((K, C_), device="cuda")
    assert lib.nsc_depthwise_bwd(xd.data_ptr(), wdd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), dwd.data_ptr(), B, C_, T, K, _st()) == 0
    assert_close(dx.cpu().numpy().transpose(0, 2, 1), xt.grad.numpy(), what="depthwise dx")
    assert_close(dwd.cpu().numpy(), wt.grad.numpy()[:, :, 0], what="depthwise dw")


def test_fast_tanh(lib):
    """nsc_tanh (csrc/nsc_common.h: Eigen's rational tanh, the arithmetic TensorFlow itself uses) is what every tanh of the
    library computes - the fused block kernels' gate, nsc_gate_fwd, nsc_apply_act.  nsc_gate_fwd writes tanh(gate) back in
    place, so it exposes the function itself: RELATIVE error against float64 tanh over the whole range (small arguments
    included: the followers' inputs are residuals), saturation, and |tanh| <= 1 (the GLU gradient uses 1 - th^2)."""
    x = np.concatenate([np.linspace(-20, 20, 20 * 8192 - 8192 - 8), np.linspace(-1e-2, 1e-2, 4096),
                        np.logspace(-30, -2, 2048), -np.logspace(-30, -2, 2048),
                        [0.0, -0.0, 1e-38, -1e-38, 88.0, -88.0, 1e30, -1e30]]).astype(np.float32)
    n = x.size // 20
    a = np.zeros((1, 40, n), np.float32)
    a[0, :20] = 1.0
    a[0, 20:] = x.reshape(20, n)
    ad = dev(a)
    g = torch.empty((1, 20, n), device="cuda")
    assert lib.nsc_gate_fwd(ad.data_ptr(), g.data_ptr(), 1, 20, n, _st()) == 0
    th = ad.cpu().numpy()[0, 20:].reshape(-1).astype(np.float64)
    ref = np.tanh(x.astype(np.float64))
    assert np.all(np.isfinite(th)) and np.all(np.abs(th) <= 1.0)
    err = np.abs(th - ref)
    bad = err > 6e-7 * np.abs(ref) + 1e-40          # (x p(x^2) passes through the denormals for |x| < 1e-35)
    assert not bad.any(), (x[bad][:5], th[bad][:5], ref[bad][:5])
    assert np.array_equal(g.cpu().numpy().reshape(-1), th.astype(np.float32))       # lin = 1: g is the tanh itself
    big = np.abs(x) > 10
    assert np.all(np.abs(th[big] - np.sign(x[big].astype(np.float64))) <= 1.2e-7)
    assert np.array_equal(np.sign(th), np.sign(x.astype(np.float64)))
    # NaN in, NaN out (the clamps of the rational form must not turn a diverged activation into -1)
    a2 = np.ones((1, 40, 64), np.float32)
    a2[0, 20:, ::3] = np.nan
    ad2 = dev(a2)
    g2 = torch.empty((1, 20, 64), device="cuda")
    assert lib.nsc_gate_fwd(ad2.data_ptr(), g2.data_ptr(), 1, 20, 64, _st()) == 0
    th2 = ad2.cpu().numpy()[0, 20:]
    assert np.all(np.isnan(th2[:, ::3])) and np.all(np.isfinite(th2[:, 1::3]))


def test_glue_kernels(lib):
    rng = np.random.default_rng(13)
    a = rng.standard_normal((2, 40, 128)).astype(np.float32)
    ad = dev(a)
    g = torch.empty((2, 20, 128), device="cuda")
    assert lib.nsc_gate_fwd(ad.data_ptr(), g.data_ptr(), 2, 20, 128, _st()) == 0
    th = np.tanh(a[:, 20:].astype(np.float64))
    assert_close(g.cpu().numpy(), a[:, :20] * th, what="gate fwd")
    dg = rng.standard_normal((2, 20, 128)).astype(np.float32)
    da = torch.empty((2, 40, 128), device="cuda")
    assert lib.nsc_gate_bwd(ad.data_ptr(), P(dg), da.data_ptr(), 2, 20, 128, _st()) == 0
    assert_close(da.cpu().numpy()[:, :20], dg * th, what="gate bwd lin")
    assert_close(da.cpu().numpy()[:, 20:], dg * a[:, :20] * (1 - th ** 2), what="gate bwd gate")
    lin, t2 = dev(a[:, :20]), dev(th)
    gg = torch.empty((2, 20, 128), device="cuda")
    assert lib.nsc_mul(lin.data_ptr(), t2.data_ptr(), gg.data_ptr(), gg.numel(), _st()) == 0
    assert_close(gg.cpu().numpy(), a[:, :20] * th, what="mul")
    dcat = torch.empty((2, 40, 128), device="cuda")
    assert lib.nsc_glu_bwd_cat(lin.data_ptr(), t2.data_ptr(), P(dg), dcat.data_ptr(), 2, 20, 128, _st()) == 0
    assert_close(dcat.cpu().numpy()[:, :20], dg * th, what="glu_cat dlin")
    assert_close(dcat.cpu().numpy()[:, 20:], dg * a[:, :20] * (1 - th ** 2), what="glu_cat dgate")
    d1, d2 = torch.empty_like(gg), torch.empty_like(gg)
    assert lib.nsc_glu_bwd(lin.data_ptr(), t2.data_ptr(), P(dg), d1.data_ptr(), d2.data_ptr(), gg.numel(), _st()) == 0
    assert_close(d1.cpu().numpy(), dg * th, what="glu dlin")
    assert_close(d2.cpu().numpy(), dg * a[:, :20] * (1 - th ** 2), what="glu dgate")
    x, y = dev(a), dev(2 * a + 1)
    out = torch.empty_like(x)
    assert lib.nsc_axpby(x.data_ptr(), y.data_ptr(), out.data_ptr(), 0.5, -2.0, x.numel(), _st()) == 0
    assert_close(out.cpu().numpy(), 0.5 * a - 2.0 * (2 * a + 1), what="axpby")
    cs = torch.ones((2, 1, 128), device="cuda")
    assert lib.nsc_channel_sum(x.data_ptr(), cs.data_ptr(), 2, 40, 128, 1, _st()) == 0
    assert_close(cs.cpu().numpy()[:, 0], 1 + a.astype(np.float64).sum(1), what="channel_sum")
    tr = torch.empty((2, 128, 40), device="cuda")
    assert lib.nsc_transpose_last2(x.data_ptr(), tr.data_ptr(), 2, 40, 128, _st()) == 0
    assert np.array_equal(tr.cpu().numpy(), a.transpose(0, 2, 1))
    s = torch.zeros(1, device="cuda")
    assert lib.nsc_sum_all(x.data_ptr(), s.data_ptr(), x.numel(), _st()) == 0
    assert abs(float(s) - a.astype(np.float64).sum()) < 1e-3 * np.abs(a).sum() ** 0.5
    idx = torch.tensor(rng.permutation(a.size).astype(np.int32), device="cuda")
    gt = torch.empty(a.size, device="cuda")
    assert lib.nsc_gather(x.data_ptr(), idx.data_ptr(), gt.data_ptr(), a.size, _st()) == 0
    assert np.array_equal(gt.cpu().numpy(), a.reshape(-1)[idx.cpu().numpy()])


QUANT_CASES = [(4, 256, 32, -20.0, True), (4, 256, 32, -300.0, True), (3, 16, 256, -50.0, True), (2, 128, 32, -20.0, False),
               (2, 37, 20, -10.0, True), (2, 64, 64, -30.0, True), (2, 9, 7, -5.0, True), (1, 16, 300, -40.0, True)]


@pytest.mark.parametrize("case", QUANT_CASES)
def test_quantizer_fwd_bwd(lib, case):
    B, L, nb, alpha, soft = case
    rng = np.random.default_rng(B * 1000 + L + nb)
    code = np.tanh(rng.standard_normal((B, L, 1))).astype(np.float32)
    bins = np.linspace(-1, 1, nb).astype(np.float32) + (0.01 * rng.standard_normal(nb)).astype(np.float32)
    on = 1.0
    c_quan, tau_scale = 10.0, 0.7 * B
    ct = torch.tensor(code, dtype=torch.float64, requires_grad=True)
    at = torch.tensor(alpha, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(bins, dtype=torch.float64, requires_grad=True)
    p, out = OT.scalar_softmax_quantization(ct, at, bt, on, soft)
    dout = rng.standard_normal((B, L, 1)).astype(np.float32)
    dp = rng.standard_normal((B, L, nb)).astype(np.float32)
    ent = OT.entropy_coding_loss(p)
    loss = (out * torch.tensor(dout, dtype=torch.float64)).sum() + (p * torch.tensor(dp, dtype=torch.float64)).sum() \
        + c_quan * OT.quan_loss(p).sum() + tau_scale * ent
    loss.backward()
    cd, ad, bd = dev(code), dev(np.array([alpha])), dev(bins)
    pd = torch.full((B, L, nb), float("nan"), device="cuda")
    od = torch.empty((B, L, 1), device="cuda")
    qd = torch.empty((B,), device="cuda")
    hist = torch.zeros((nb,), device="cuda")
    rc = lib.nsc_quantize_fwd(cd.data_ptr(), ad.data_ptr(), bd.data_ptr(), on, int(soft), B, L, nb, pd.data_ptr(),
                              od.data_ptr(), qd.data_ptr(), hist.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    assert_close(pd.cpu().numpy(), p.detach().numpy(), what="p")
    assert_close(od.cpu().numpy(), out.detach().numpy(), what="out")
    assert_close(qd.cpu().numpy(), OT.quan_loss(p).detach().numpy(), what="quan partial")
    assert_close(hist.cpu().numpy(), p.detach().numpy().reshape(-1, nb).sum(0), what="hist")
    e = torch.empty(1, device="cuda"); gh = torch.empty(nb, device="cuda")
    assert lib.nsc_entropy_from_hist(hist.data_ptr(), nb, e.data_ptr(), gh.data_ptr(), _st()) == 0
    assert abs(float(e) - float(ent)) < 1e-4 * max(1.0, abs(float(ent)))
    dc = torch.full((B, L, 1), float("nan"), device="cuda")
    da = torch.zeros(1, device="cuda"); db = torch.zeros(nb, device="cuda")
    rc = lib.nsc_quantize_bwd(cd.data_ptr(), ad.data_ptr(), bd.data_ptr(), on, int(soft), B, L, nb, P(dout),
                              P(dp), c_quan, gh.data_ptr(), tau_scale, 0, dc.data_ptr(), da.data_ptr(),
                              db.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    assert_close(dc.cpu().numpy(), ct.grad.numpy(), tol=2e-4, what="dcode")
    assert_close(db.cpu().numpy(), bt.grad.numpy(), tol=2e-4, what="dbins")
    assert abs(float(da) - float(at.grad)) <= 2e-4 * max(abs(float(at.grad)), 1e-3), ("dalpha", float(da), float(at.grad))


@pytest.mark.parametrize("L,soft", [(256, True), (128, True), (256, False)])
def test_quantizer_large_batch_wave_per_frame_kernel(lib, L, soft):
    """At B >= 1024 with p materialised and 32 bins nsc_quantize_fwd runs the wave-per-frame kernel (csrc/quant.hip):
    p and the soft codes must equal the workgroup-per-frame kernel's (same frames in launches of <= 512) up to rounding - since
    round 4 the wave kernel evaluates p_k = t_k^2 / sum t^2 with t = 2^(ah (min |d| - |d_k|)), one transcendental per bin, the
    workgroup kernel exp / rcp: ~3 ulp apart -, HARD codes bit for bit, quan_loss / histogram up to rounding and summation
    order, a slice of frames is held to the float64 oracle, and frames whose code sits exactly between two bins pick the lower
    index (tf.nn.top_k) in hard mode."""
    B, nb = 1030, 32
    rng = np.random.default_rng(L + int(soft))
    code = np.tanh(rng.standard_normal((B, L, 1))).astype(np.float32)
    bins = ((np.arange(nb) - 15.5) / 16.0).astype(np.float32)          # dyadic bins: midpoints and distances are exact in fp32
    code[7, :nb - 1, 0] = 0.5 * (bins[:-1] + bins[1:])                 # codes exactly between two bins
    alpha = -20.0
    cd, ad, bd = dev(code), dev(np.array([alpha])), dev(bins)

    def run(lo, hi):
        n = hi - lo
        pd = torch.full((n, L, nb), float("nan"), device="cuda")
        od = torch.full((n, L, 1), float("nan"), device="cuda")
        qd = torch.full((n,), float("nan"), device="cuda")
        hist = torch.zeros((nb,), device="cuda")
        assert lib.nsc_quantize_fwd(cd[lo:hi].contiguous().data_ptr(), ad.data_ptr(), bd.data_ptr(), 1.0, int(soft), n, L, nb,
                                    pd.data_ptr(), od.data_ptr(), qd.data_ptr(), hist.data_ptr(), _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        return pd.cpu().numpy(), od.cpu().numpy(), qd.cpu().numpy(), hist.cpu().numpy()

    big = run(0, B)                                                     # one launch: wave per frame
    parts = [run(0, 512), run(512, 1024), run(1024, B)]                 # B < 1024 each: workgroup per frame
    assert_close(big[0], np.concatenate([p_[0] for p_ in parts], 0), tol=2e-6, atol=1e-9, what="p, wave vs workgroup kernel")
    if soft:
        assert_close(big[1], np.concatenate([p_[1] for p_ in parts], 0), tol=2e-6, atol=1e-7, what="soft codes, wave vs workgroup kernel")
    else:
        assert np.array_equal(big[1], np.concatenate([p_[1] for p_ in parts], 0)), "hard codes"
    assert_close(big[2], np.concatenate([p_[2] for p_ in parts], 0), tol=2e-6, what="quan partial")
    assert_close(big[3], sum(p_[3] for p_ in parts), tol=1e-5, what="histogram")
    sl = slice(0, 16)
    pt, ot = OT.scalar_softmax_quantization(torch.tensor(code[sl], dtype=torch.float64), torch.tensor(alpha, dtype=torch.float64),
                                            torch.tensor(bins, dtype=torch.float64), 1.0, soft)
    assert_close(big[0][sl], pt.numpy(), what="p vs oracle", atol=1e-7)
    if soft:
        assert_close(big[1][sl], ot.numpy(), what="codes vs oracle", atol=1e-6)
    else:
        assert np.array_equal(big[1][7, :nb - 1, 0], bins[:-1])          # ties -> lowest index, exact bin values


def test_quantizer_identity_and_nearest_bin(lib):
    rng = np.random.default_rng(2)
    B, L, nb = 2, 256, 32
    code = rng.uniform(-0.99, 0.99, (B, L, 1)).astype(np.float32)
    bins = np.linspace(-1, 1, nb).astype(np.float32)
    cd, ad, bd = dev(code), dev(np.array([-300.0])), dev(bins)
    od = torch.empty((B, L, 1), device="cuda")
    assert lib.nsc_quantize_fwd(cd.data_ptr(), ad.data_ptr(), bd.data_ptr(), 0.0, 1, B, L, nb, None, od.data_ptr(), None, None, _st()) == 0
    assert np.array_equal(od.cpu().numpy(), code)                       # is_quan_on = 0 -> identity, bit exact
    assert lib.nsc_quantize_fwd(cd.data_ptr(), ad.data_ptr(), bd.data_ptr(), 1.0, 0, B, L, nb, None, od.data_ptr(), None, None, _st()) == 0
    idx = np.argmin(np.abs(code.astype(np.float64) - bins.astype(np.float64)), axis=-1)
    assert np.array_equal(od.cpu().numpy()[..., 0], bins[idx])          # hard: exact bin values, exact indices
    # tie -> lowest index
    assert lib.nsc_quantize_fwd(P(np.array([[[0.5]]])), ad.data_ptr(), P(np.array([0.0, 1.0, 2.0, 3.0])),
                                1.0, 0, 1, 1, 4, None, od.data_ptr(), None, None, _st()) == 0
    assert float(od.reshape(-1)[0]) == 0.0


def test_recon_loss_and_rfft(lib):
    from nsc_amd.loss_terms_and_measures import mel_matrix_cat
    rng = np.random.default_rng(17)
    B = 5
    tgt = (0.03 * rng.standard_normal((B, 512))).astype(np.float32)
    dec = (tgt + 0.01 * rng.standard_normal((B, 512))).astype(np.float32)
    dec[0] = tgt[0]                                                      # zero-error frame: eps paths
    dt = torch.tensor(dec, dtype=torch.float64, requires_grad=True)
    tt = torch.tensor(tgt, dtype=torch.float64)
    tl, fl = OT.mse_loss(dt, tt), OT.mfcc_loss(dt, tt)
    (60.0 * tl + 10.0 * fl).sum().backward()
    mel = mel_matrix_cat()
    md, mtd = dev(mel), dev(np.ascontiguousarray(mel.T))
    to, fo = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    g = torch.empty((B, 512), device="cuda")
    rc = lib.nsc_recon_loss(P(dec), P(tgt), B, 60.0, 10.0, None, None, md.data_ptr(),
                            mtd.data_ptr(), to.data_ptr(), fo.data_ptr(), g.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    assert_close(to.cpu().numpy(), tl.detach().numpy(), what="time loss")
    assert_close(fo.cpu().numpy(), fl.detach().numpy(), what="freq loss")
    assert_close(g.cpu().numpy()[1:], dt.grad.numpy()[1:], tol=5e-4, what="recon grad")
    # banded form (the engine's): only the non-zero band of the mel matrix is visited -> the very same floats
    from nsc_amd.loss_terms_and_measures import mel_band_ranges
    rg = torch.from_numpy(mel_band_ranges()).cuda()
    to2, fo2, g2 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty((B, 512), device="cuda")
    rc = lib.nsc_recon_loss_banded(P(dec), P(tgt), B, 60.0, 10.0, None, None, md.data_ptr(), mtd.data_ptr(), rg.data_ptr(),
                                   to2.data_ptr(), fo2.data_ptr(), g2.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    assert torch.equal(to2, to) and torch.equal(fo2, fo) and torch.equal(g2[1:], g[1:])
    # frame 0 (decoded == target): its gradient is rounding noise of the two spectra divided by ~sqrt(1e-7); last-bit level
    assert float((g2[0] - g[0]).abs().max()) <= 1e-6 * float(g[0].abs().max())
    # bare rFFT + cosine KAT
    sig = np.stack([np.cos(2 * np.pi * 5 * np.arange(512) / 512), rng.standard_normal(512)]).astype(np.float32)
    re, im, mag = (torch.empty((2, 257), device="cuda") for _ in range(3))
    assert lib.nsc_rfft512(P(sig), 2, re.data_ptr(), im.data_ptr(), mag.data_ptr(), _st()) == 0
    st, m = O.tf_stft(sig)
    assert abs(float(mag[0, 5]) - 256.0) < 1e-3
    assert_close(re.cpu().numpy(), st.real, what="rfft re")
    assert_close(im.cpu().numpy(), st.imag, what="rfft im")
    assert_close(mag.cpu().numpy(), m, what="rfft mag")


def test_adam_tf1(lib):
    rng = np.random.default_rng(19)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32)
    pd = dev(p); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        g[::7] = 0.0
        assert lib.nsc_adam_tf1_step(pd.data_ptr(), P(g), m.data_ptr(), v.data_ptr(), n, 2e-4, 0.9, 0.999, 1e-8, t, None, _st()) == 0
        pr, mr, vr = O.adam_tf1_step(pr, g.astype(np.float64), mr, vr, t, 2e-4)
    assert np.max(np.abs(pd.cpu().numpy() - pr)) < 1e-6   # p ~ N(0,1): a few fp32 ulps after 3 steps
    z = dev(np.ones(4)); mz = torch.zeros(4, device="cuda"); vz = torch.zeros(4, device="cuda")
    assert lib.nsc_adam_tf1_step(z.data_ptr(), torch.zeros(4, device="cuda").data_ptr(), mz.data_ptr(), vz.data_ptr(), 4, 1e-3, 0.9, 0.999, 1e-8, 1, None, _st()) == 0
    assert np.array_equal(z.cpu().numpy(), np.ones(4, np.float32))      # zero gradient must not move a variable


def test_framing_and_overlap_add(lib):
    rng = np.random.default_rng(23)
    for n in (513, 993, 2000, 16000):
        utt = rng.standard_normal(n).astype(np.float32)
        ref = O.utterance_to_segment(utt.astype(np.float64), True)
        nf = ref.shape[0]
        fr = torch.empty((nf, 512), device="cuda")
        assert lib.nsc_frame_utterance(P(utt), n, None, fr.data_ptr(), nf, _st()) == 0
        assert np.array_equal(fr.cpu().numpy(), ref.astype(np.float32))   # bit-exact frame indexing
        win = O.training_window().astype(np.float32)
        assert lib.nsc_frame_utterance(P(utt), n, P(win), fr.data_ptr(), nf, _st()) == 0
        assert np.array_equal(fr.cpu().numpy(), utt_frames_win(utt, win, nf))
        win3 = np.stack([O.hann_process(np.ones(512), 0, 3), O.hann_process(np.ones(512), 1, 3), O.hann_process(np.ones(512), 2, 3)]).astype(np.float32)
        out = torch.empty(480 * (nf - 1) + 512, device="cuda")
        frames = rng.standard_normal((nf, 512)).astype(np.float32)
        assert lib.nsc_overlap_add(P(frames), nf, P(win3), out.data_ptr(), _st()) == 0
        assert_close(out.cpu().numpy(), O.overlap_add(frames.astype(np.float64)), tol=1e-6, what=f"overlap_add n={n}")


def utt_frames_win(utt, win, nf):
    return np.stack([utt[480 * i:480 * i + 512] * win for i in range(nf)]).astype(np.float32)


def test_bad_arguments_return_status(lib):
    d = _desc(B=0)
    assert lib.nsc_conv1d_fwd(C.byref(d), None, None, None, None, None, None, None) == -1
    assert b"desc" in lib.nsc_last_error() or b"size" in lib.nsc_last_error()
    assert lib.nsc_quantize_fwd(None, None, None, 1.0, 1, 1, 1, 1, None, None, None, None, None) == -1
    assert lib.nsc_adam_tf1_step(None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 1, None, None) == -1


@pytest.mark.parametrize("case", [(2, 100, 512, 2, 0), (2, 100, 256, 1, 1), (3, 50, 512, 2, 0), (2, 50, 512, 1, 1),
                                  (1, 100, 200, 2, 0), (2, 36, 70, 1, 0), (70, 100, 300, 2, 0), (40, 50, 512, 1, 1),
                                  # long chains of consecutive tiles (carried h / g columns), crossing frame boundaries
                                  (150, 100, 512, 1, 0), (150, 50, 512, 2, 1), (72, 100, 256, 2, 0),
                                  # C = 25 (third resolution of '2 2' codecs): persistent kernel on the C = 50 job table
                                  (3, 25, 128, 1, 0), (300, 25, 128, 2, 1), (2, 25, 200, 2, 0),
                                  # T not a multiple of 4: rows are not 16-byte aligned (element-wise masks, scalar stores)
                                  (3, 100, 130, 2, 0), (40, 100, 203, 1, 1), (5, 50, 70, 2, 0), (3, 25, 67, 1, 0)])
def test_fused_gated_block_fwd(lib, case):
    """csrc/block.hip vs the oracle's gated_bottleneck (nn_core_operator.py:82-112), incl. saved intermediates."""
    B, C_, T, dil, flat = case
    rng = np.random.default_rng(100 + C_ + T + dil)
    ps = O.ParamStore(rng)
    x = rng.standard_normal((B, T, C_)).astype(np.float32)
    tape = []
    ref = O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, bool(flat), tape)
    names = ["s/conv1d", "s/conv1d_1", "s/conv1d_2", "s/conv1d_3"]
    for n in names:
        ps.params[n + "/bias"] = (0.1 * rng.standard_normal(ps.params[n + "/bias"].shape)).astype(np.float32).astype(np.float64)
    ps.begin_replay()
    tape = []
    ref = O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, bool(flat), tape)
    t = tape[0][1]
    W = [P(ps.params[n + "/kernel"]) for n in names]
    Bv = [P(ps.params[n + "/bias"]) for n in names]
    out = torch.full((B, C_, T), float("nan"), device="cuda")
    h, lin, th, g = (torch.full((B, 20, T), float("nan"), device="cuda") for _ in range(4))
    rc = lib.nsc_gated_block_fwd(P(x.transpose(0, 2, 1)), W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3],
                                 out.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), g.data_ptr(), B, C_, T,
                                 20, 9, dil, flat, _st())
    assert rc == 0, lib.nsc_last_error()
    tr = lambda v: v.cpu().numpy().transpose(0, 2, 1)
    assert_close(tr(h), t["h"], what="fused h")
    assert_close(tr(lin), t["left"], what="fused lin")
    assert_close(tr(th), t["right"], what="fused tanh branch")
    assert_close(tr(g), t["g"], what="fused g")
    assert_close(tr(out), ref, what=f"fused block out {case}")
    # without the optional outputs
    out2 = torch.full((B, C_, T), float("nan"), device="cuda")
    assert lib.nsc_gated_block_fwd(P(x.transpose(0, 2, 1)), W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3],
                                   out2.data_ptr(), None, None, None, None, B, C_, T, 20, 9, dil, flat, _st()) == 0
    assert torch.equal(out, out2)
    # an output tensor that is only 4-byte aligned (a view one float into a buffer): the 16-byte row stores of phase 3 give way
    # to the per-element form, same values
    buf = torch.full((B * C_ * T + 1,), float("nan"), device="cuda")
    out3 = buf[1:].view(B, C_, T)
    assert lib.nsc_gated_block_fwd(P(x.transpose(0, 2, 1)), W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3],
                                   out3.data_ptr(), None, None, None, None, B, C_, T, 20, 9, dil, flat, _st()) == 0
    assert torch.equal(out, out3) and bool(torch.isnan(buf[0]))


@pytest.mark.parametrize("case", [(2, 100, 512, 2), (2, 100, 256, 1), (3, 50, 512, 2), (5, 36, 70, 1)])
def test_block_wgrad_kernel(lib, case):
    """Persistent block weight-gradient kernel vs per-conv oracle gradients (inputs are arbitrary tensors)."""
    B, C_, T, dil = case
    rng = np.random.default_rng(500 + C_ + T + dil)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    x, dy = f(B, T, C_), f(B, T, C_)
    h, g, dlin, dgate, dz1 = (f(B, T, 20) for _ in range(5))
    def wg(inp, grad, K, dil_):
        it = torch.tensor(inp, dtype=torch.float64)
        W = torch.zeros((K, inp.shape[-1], grad.shape[-1]), dtype=torch.float64, requires_grad=True)
        bb = torch.zeros(grad.shape[-1], dtype=torch.float64, requires_grad=True)
        (OT.conv1d(it, W, bb, dil_, 1, None) * torch.tensor(grad, dtype=torch.float64)).sum().backward()
        return W.grad.numpy(), bb.grad.numpy()
    ref = [wg(x, dz1, 1, 1), wg(h, dlin, 15, dil), wg(h, dgate, 15, dil), wg(g, dy, 9, 1)]
    shapes = [(1, C_, 20), (15, 20, 20), (15, 20, 20), (9, 20, C_)]
    dws = [torch.zeros(s, device="cuda") for s in shapes]
    dbs = [torch.zeros(s[2], device="cuda") for s in shapes]
    tr = lambda v: P(v.transpose(0, 2, 1))
    da = np.concatenate([dlin, dgate], axis=2)                              # [B,T,40] -> dlin | dgate
    rc = lib.nsc_gated_block_wgrad(tr(x), tr(h), tr(g), tr(dy), tr(da), tr(dz1), dws[0].data_ptr(), dbs[0].data_ptr(),
                                   dws[1].data_ptr(), dbs[1].data_ptr(), dws[2].data_ptr(), dbs[2].data_ptr(),
                                   dws[3].data_ptr(), dbs[3].data_ptr(), None, None, 0, B, C_, T, 20, 9, dil, 4, 1, None, _st())
    assert rc == 0, lib.nsc_last_error()
    rc = lib.nsc_gated_block_wgrad(tr(x), tr(h), tr(g), tr(dy), tr(da), tr(dz1), dws[0].data_ptr(), dbs[0].data_ptr(),
                                   dws[1].data_ptr(), dbs[1].data_ptr(), dws[2].data_ptr(), dbs[2].data_ptr(),
                                   dws[3].data_ptr(), dbs[3].data_ptr(), None, None, 0, B, C_, T, 20, 9, dil, 4, 2, None, _st())
    assert rc == 0, lib.nsc_last_error()
    for i in range(4):
        assert_close(dws[i].cpu().numpy(), ref[i][0], tol=2e-4, what=f"block wgrad dW[{i}] {case}")
        assert_close(dbs[i].cpu().numpy(), ref[i][1], tol=2e-4, what=f"block wgrad db[{i}] {case}")
    # store + reduce flush (gradients contiguous in one flat buffer, pre-filled to check accumulation) together with
    # the fused 1x1 data gradient dx = (W1^T dz1 + dy) * lrelu'(x)
    sizes = [int(np.prod(s)) for s in shapes]
    flat = torch.full((sum(sizes) + sum(s[2] for s in shapes),), 0.5, device="cuda")
    ptrs, off = [], 0
    for s_, n_ in zip(shapes, sizes):
        ptrs += [flat.data_ptr() + 4 * off, flat.data_ptr() + 4 * (off + n_)]
        off += n_ + s_[2]
    ws = torch.empty(int(lib.nsc_gated_block_wgrad_workspace(C_)), device="cuda")
    W1 = (rng.standard_normal((1, C_, 20)) / 5).astype(np.float32)
    wt1 = np.ascontiguousarray(W1[::-1].transpose(0, 2, 1))                  # [1,20,C]
    dx = torch.full((B, C_, T), float("nan"), device="cuda")
    rc = lib.nsc_gated_block_wgrad(tr(x), tr(h), tr(g), tr(dy), tr(da), tr(dz1), *ptrs, P(wt1), dx.data_ptr(), 2, B, C_, T,
                                   20, 9, dil, 8, 0, ws.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    got, off = flat.cpu().numpy() - 0.5, 0
    for i, (s_, n_) in enumerate(zip(shapes, sizes)):
        assert_close(got[off:off + n_].reshape(s_), ref[i][0], tol=2e-4, what=f"slab flush dW[{i}] {case}")
        assert_close(got[off + n_:off + n_ + s_[2]], ref[i][1], tol=2e-4, what=f"slab flush db[{i}] {case}")
        off += n_ + s_[2]
    dx_ref = (dz1.astype(np.float64) @ W1[0].astype(np.float64).T + dy) * np.where(x > 0, 1.0, 0.2)
    assert_close(dx.cpu().numpy().transpose(0, 2, 1), dx_ref, tol=2e-4, what=f"fused 1x1 dgrad {case}")
    # the two light parts with the slab flush (each must reduce only what it wrote; workspace poisoned first)
    flat.fill_(0.25)
    ws.fill_(float("nan"))
    for part in (1, 2):
        rc = lib.nsc_gated_block_wgrad(tr(x), tr(h), tr(g), tr(dy), tr(da), tr(dz1), *ptrs, None, None, 0, B, C_, T, 20, 9, dil,
                                       4, part, ws.data_ptr(), _st())
        assert rc == 0, lib.nsc_last_error()
    got, off = flat.cpu().numpy() - 0.25, 0
    for i, (s_, n_) in enumerate(zip(shapes, sizes)):
        assert_close(got[off:off + n_].reshape(s_), ref[i][0], tol=2e-4, what=f"split slab flush dW[{i}] {case}")
        assert_close(got[off + n_:off + n_ + s_[2]], ref[i][1], tol=2e-4, what=f"split slab flush db[{i}] {case}")
        off += n_ + s_[2]


@pytest.mark.parametrize("case", [(2, 100, 256, 1, 0), (3, 100, 256, 2, 1), (2, 50, 512, 1, 0), (150, 100, 256, 2, 0), (70, 100, 300, 1, 0),
                                  (3, 25, 128, 1, 0), (300, 25, 128, 2, 1), (3, 100, 130, 2, 0), (40, 50, 203, 1, 0)])
def test_fused_gated_block_fwd_one_input_channel(lib, case):
    """nsc_gated_block_fwd_cin1 vs the oracle's gated_bottleneck on a [B,T,1] input (broadcast residual), incl. the saved
    intermediates and chains of tiles."""
    B, C_, T, dil, flat = case
    rng = np.random.default_rng(900 + C_ + T + dil)
    ps = O.ParamStore(rng)
    x = rng.standard_normal((B, T, 1)).astype(np.float32)
    O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, bool(flat))
    names = ["s/conv1d", "s/conv1d_1", "s/conv1d_2", "s/conv1d_3"]
    for n in names:
        ps.params[n + "/bias"] = (0.1 * rng.standard_normal(ps.params[n + "/bias"].shape)).astype(np.float32).astype(np.float64)
    ps.begin_replay()
    tape = []
    ref = O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, bool(flat), tape)
    t = tape[0][1]
    W = [P(ps.params[n + "/kernel"]) for n in names]
    Bv = [P(ps.params[n + "/bias"]) for n in names]
    out = torch.full((B, C_, T), float("nan"), device="cuda")
    h, lin, th, g = (torch.full((B, 20, T), float("nan"), device="cuda") for _ in range(4))
    rc = lib.nsc_gated_block_fwd_cin1(P(x.transpose(0, 2, 1)), W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3],
                                      out.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), g.data_ptr(), B, C_, T,
                                      20, 9, dil, flat, _st())
    assert rc == 0, lib.nsc_last_error()
    tr = lambda v: v.cpu().numpy().transpose(0, 2, 1)
    assert_close(tr(h), t["h"], what="cin1 h")
    assert_close(tr(lin), t["left"], what="cin1 lin")
    assert_close(tr(th), t["right"], what="cin1 tanh branch")
    assert_close(tr(g), t["g"], what="cin1 g")
    assert_close(tr(out), ref, what=f"cin1 block out {case}")
    assert lib.nsc_gated_block_fwd_cin1(P(x.transpose(0, 2, 1)), W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3],
                                        out.data_ptr(), None, None, None, None, B, 36, T, 20, 9, dil, flat, _st()) == -2


_POISON = []


def _poison_lds_with_nan(lib):
    """Leaves NaN in the LDS of every CU: one C = 100 data-gradient launch on all-NaN tensors and weights (its tiles and weight
    tables cover the 160 KB).  A kernel that then reads LDS it never wrote - a table pad, a channel group past C - picks the
    NaN up (NaN x 0 = NaN), instead of passing by the luck of what the previous kernel left there."""
    B, C_, T = 256, 100, 64
    if not _POISON:
        nan = lambda *sh: torch.full(sh, float("nan"), device="cuda")
        _POISON.extend([nan(B, C_, T), nan(B, C_, T), nan(B, 20, T), nan(B, 20, T), nan(B, 20, T), nan(1, 20, C_), nan(15, 20, 20),
                        nan(15, 20, 20), nan(9, C_, 20), nan(B, C_, T), nan(B, 40, T), nan(B, 20, T)])
    x, dy, h, lin, th, w1, wl, wr, w9, dx, da, dz1 = _POISON
    assert lib.nsc_gated_block_dgrad(x.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), dy.data_ptr(), w1.data_ptr(),
                                     wl.data_ptr(), wr.data_ptr(), w9.data_ptr(), dx.data_ptr(), da.data_ptr(), dz1.data_ptr(),
                                     B, C_, T, 20, 9, 2, 0, _st()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(2, 100, 512, 2, 2), (2, 100, 256, 1, 0), (3, 50, 512, 2, 2), (2, 50, 512, 1, 2),
                                  (1, 100, 200, 2, 0), (5, 36, 70, 1, 2),
                                  # more tiles than workgroups: chains of consecutive tiles with the carried da halo, chains
                                  # that cross frame boundaries, a ragged last tile
                                  (40, 100, 512, 2, 2), (72, 100, 256, 1, 2), (150, 50, 512, 2, 2), (61, 100, 300, 1, 0),
                                  # C = 25: dy staged as 36 rows (9 channel groups) on the C = 50 job table
                                  (3, 25, 128, 1, 2), (300, 25, 128, 2, 2), (2, 25, 200, 2, 0),
                                  # T not a multiple of 4: rows are not 16-byte aligned (element-wise masks, scalar stores)
                                  (3, 100, 130, 2, 2), (40, 100, 203, 1, 2), (5, 50, 70, 2, 0), (3, 25, 67, 1, 2)])
def test_fused_gated_block_dgrad(lib, case):
    """8-wave data-path backward (dx, dlin|dgate, dz1) vs autograd of the oracle block with its saved intermediates."""
    B, C_, T, dil, in_act = case
    rng = np.random.default_rng(700 + C_ + T + dil)
    names = ["s/conv1d", "s/conv1d_1", "s/conv1d_2", "s/conv1d_3"]
    ps = O.ParamStore(rng)
    x = rng.standard_normal((B, T, C_)).astype(np.float32)
    O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, True)
    for n in names:
        ps.params[n + "/bias"] = (0.1 * rng.standard_normal(ps.params[n + "/bias"].shape)).astype(np.float32).astype(np.float64)
    dy = rng.standard_normal((B, T, C_)).astype(np.float32)
    tp = OT.TorchParams(ps)
    zt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    xin = torch.nn.functional.leaky_relu(zt, 0.2) if in_act == 2 else zt
    # forward with retained intermediates (same graph as OT.gated_bottleneck)
    W1, b1 = tp.conv("s"); Wl, bl = tp.conv("s"); Wr, br = tp.conv("s"); W9, b9 = tp.conv("s")
    hpre = OT.conv1d(xin, W1, b1, activation=None); hpre.retain_grad()
    h = OT.act(hpre, "lrelu")
    left = OT.conv1d(h, Wl, bl, dilation_rate=dil, activation=None); left.retain_grad()
    rpre = OT.conv1d(h, Wr, br, dilation_rate=dil, activation=None); rpre.retain_grad()
    right = torch.tanh(rpre)
    y = OT.conv1d(left * right, W9, b9, activation=None) + xin
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    xin_np = xin.detach().numpy().astype(np.float32)
    wt = {n: np.ascontiguousarray(ps.params[n + "/kernel"].astype(np.float32)[::-1].transpose(0, 2, 1)) for n in names}
    tr = lambda v: P(np.ascontiguousarray(np.asarray(v, np.float32).transpose(0, 2, 1)))
    dx = torch.full((B, C_, T), float("nan"), device="cuda")
    da = torch.full((B, 40, T), float("nan"), device="cuda")
    dz1 = torch.full((B, 20, T), float("nan"), device="cuda")
    rc = lib.nsc_gated_block_dgrad(tr(xin_np), tr(h.detach().numpy()), tr(left.detach().numpy()), tr(right.detach().numpy()),
                                   tr(dy), P(wt[names[0]]), P(wt[names[1]]), P(wt[names[2]]), P(wt[names[3]]), dx.data_ptr(),
                                   da.data_ptr(), dz1.data_ptr(), B, C_, T, 20, 9, dil, in_act, _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    g = lambda v: v.cpu().numpy().transpose(0, 2, 1)
    assert_close(g(da)[:, :, :20], left.grad.numpy(), tol=2e-4, what=f"dlin {case}")
    assert_close(g(da)[:, :, 20:], rpre.grad.numpy(), tol=2e-4, what=f"dgate {case}")
    assert_close(g(dz1), hpre.grad.numpy(), tol=2e-4, what=f"dz1 {case}")
    assert_close(g(dx), zt.grad.numpy(), tol=2e-4, what=f"dx {case}")


@pytest.mark.parametrize("case", [(2, 100, 256, 1), (3, 100, 256, 2), (2, 50, 512, 2), (150, 100, 256, 2), (70, 100, 300, 1),
                                  (3, 25, 128, 1), (300, 25, 128, 2), (3, 100, 130, 2), (40, 50, 203, 1)])
def test_fused_gated_block_dgrad_one_input_channel(lib, case):
    """nsc_gated_block_dgrad_cin1 vs autograd of the oracle block on a [B,T,1] input: dx (incl. the channel-summed residual
    branch), dlin, dgate, dz1 - with chains of tiles."""
    B, C_, T, dil = case
    rng = np.random.default_rng(1100 + C_ + T + dil)
    names = ["s/conv1d", "s/conv1d_1", "s/conv1d_2", "s/conv1d_3"]
    ps = O.ParamStore(rng)
    x = rng.standard_normal((B, T, 1)).astype(np.float32)
    O.gated_bottleneck(x, ps, "s", C_, 20, 9, dil, True)
    for n in names:
        ps.params[n + "/bias"] = (0.1 * rng.standard_normal(ps.params[n + "/bias"].shape)).astype(np.float32).astype(np.float64)
    dy = rng.standard_normal((B, T, C_)).astype(np.float32)
    tp = OT.TorchParams(ps)
    xin = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    W1, b1 = tp.conv("s"); Wl, bl = tp.conv("s"); Wr, br = tp.conv("s"); W9, b9 = tp.conv("s")
    hpre = OT.conv1d(xin, W1, b1, activation=None); hpre.retain_grad()
    h = OT.act(hpre, "lrelu")
    left = OT.conv1d(h, Wl, bl, dilation_rate=dil, activation=None); left.retain_grad()
    rpre = OT.conv1d(h, Wr, br, dilation_rate=dil, activation=None); rpre.retain_grad()
    right = torch.tanh(rpre)
    y = OT.conv1d(left * right, W9, b9, activation=None) + xin          # [B,T,C] + [B,T,1]: broadcast
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    wt = {n: np.array(ps.params[n + "/kernel"].astype(np.float32)[::-1].transpose(0, 2, 1), order="C", copy=True) for n in names}
    tr = lambda v: P(np.ascontiguousarray(np.asarray(v, np.float32).transpose(0, 2, 1)))
    dx = torch.full((B, 1, T), float("nan"), device="cuda")
    dlin, dgate, dz1 = (torch.full((B, 20, T), float("nan"), device="cuda") for _ in range(3))
    rc = lib.nsc_gated_block_dgrad_cin1(tr(h.detach().numpy()), tr(left.detach().numpy()), tr(right.detach().numpy()), tr(dy),
                                        P(wt[names[0]]), P(wt[names[1]]), P(wt[names[2]]), P(wt[names[3]]), dx.data_ptr(),
                                        dlin.data_ptr(), dgate.data_ptr(), dz1.data_ptr(), B, C_, T, 20, 9, dil, 20, _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    g = lambda v: v.cpu().numpy().transpose(0, 2, 1)
    assert_close(g(dlin), left.grad.numpy(), tol=2e-4, what=f"cin1 dlin {case}")
    assert_close(g(dgate), rpre.grad.numpy(), tol=2e-4, what=f"cin1 dgate {case}")
    assert_close(g(dz1), hpre.grad.numpy(), tol=2e-4, what=f"cin1 dz1 {case}")
    assert_close(g(dx), xin.grad.numpy(), tol=2e-4, what=f"cin1 dx {case}")


def test_block_wgrad_batch_equals_per_block_launches(lib):
    """nsc_gated_block_wgrad_batch (deferred, one launch for many blocks of mixed shapes) == per-block nsc_gated_block_wgrad."""
    from nsc_amd._lib import BlockWgradJob
    rng = np.random.default_rng(77)
    B = 6
    shapes = [(100, 256, 2), (100, 512, 1), (50, 512, 2), (100, 256, 1), (50, 512, 1), (36, 128, 2)]
    keep, jobs, refs, outs = [], [], [], []
    lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
    nws = lib.nsc_gated_block_wgrad_batch_workspace(100)
    ws = torch.full((nws,), float("nan"), device="cuda")
    for (C_, T, dil) in shapes:
        t = {k: dev(rng.standard_normal(s_).astype(np.float32)) for k, s_ in
             dict(x=(B, C_, T), h=(B, 20, T), g=(B, 20, T), dy=(B, C_, T), da=(B, 40, T), dz1=(B, 20, T)).items()}
        n = C_ * 20 + 20 + 2 * (15 * 20 * 20 + 20) + 9 * 20 * C_ + C_
        ref = torch.full((n,), 0.5, device="cuda")
        out = torch.full((n,), 0.5, device="cuda")
        o = [0, C_ * 20, C_ * 20 + 20, C_ * 20 + 20 + 6000, C_ * 20 + 20 + 6020, C_ * 20 + 20 + 12020, C_ * 20 + 20 + 12040,
             C_ * 20 + 20 + 12040 + 180 * C_]
        ptrs = [ref.data_ptr() + 4 * v for v in o]
        rc = lib.nsc_gated_block_wgrad(t["x"].data_ptr(), t["h"].data_ptr(), t["g"].data_ptr(), t["dy"].data_ptr(),
                                       t["da"].data_ptr(), t["dz1"].data_ptr(), *ptrs, None, None, 0, B, C_, T, 20, 9, dil, 8, 0,
                                       None, _st())
        assert rc == 0, lib.nsc_last_error()
        jobs.append(BlockWgradJob(t["x"].data_ptr(), t["h"].data_ptr(), t["g"].data_ptr(), t["dy"].data_ptr(), t["da"].data_ptr(),
                                  t["dz1"].data_ptr(), out.data_ptr(), C_, T, dil))
        keep.append(t); refs.append(ref); outs.append(out)
    arr = (BlockWgradJob * len(jobs))(*jobs)
    rc = lib.nsc_gated_block_wgrad_batch(arr, len(jobs), B, 20, 9, ws.data_ptr(), nws, _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for i, (r, o_) in enumerate(zip(refs, outs)):
        assert_close(o_.cpu().numpy(), r.cpu().numpy(), tol=2e-4, what=f"batched block wgrad job {i} {shapes[i]}")


def test_block_wgrad_batch_one_input_channel_jobs(lib):
    """Jobs with Cin = 1 (first block of a decoder stage; dW1 is [1,20], x one row) in the same batched launch as C -> C
    jobs, against float64 autograd of the four convolutions."""
    import torch.nn.functional as F
    from nsc_amd._lib import BlockWgradJob
    rng = np.random.default_rng(78)
    B = 5
    shapes = [(100, 1, 512, 1), (50, 50, 256, 2), (50, 1, 256, 2), (100, 1, 128, 2)]      # (C, Cin, T, dil)
    lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
    nws = lib.nsc_gated_block_wgrad_batch_workspace(100)
    ws = torch.full((nws,), float("nan"), device="cuda")
    keep, jobs, outs, wants = [], [], [], []
    for (C_, Ci, T, dil) in shapes:
        t = {k: dev(rng.standard_normal(s_).astype(np.float32)) for k, s_ in
             dict(x=(B, Ci, T), h=(B, 20, T), g=(B, 20, T), dy=(B, C_, T), da=(B, 40, T), dz1=(B, 20, T)).items()}
        d = {k: v.double() for k, v in t.items()}
        def wgrad(inp, dout, K, dl):
            w = torch.zeros(dout.shape[1], inp.shape[1], K, dtype=torch.float64, device="cuda", requires_grad=True)
            b = torch.zeros(dout.shape[1], dtype=torch.float64, device="cuda", requires_grad=True)
            (F.conv1d(inp, w, b, dilation=dl, padding=(K - 1) // 2 * dl) * dout).sum().backward()
            return [w.grad.permute(2, 1, 0).reshape(-1), b.grad]          # kernel layout [K, Cin, Cout]
        want = torch.cat(wgrad(d["x"], d["dz1"], 1, 1) + wgrad(d["h"], d["da"][:, :20], 15, dil) +
                         wgrad(d["h"], d["da"][:, 20:], 15, dil) + wgrad(d["g"], d["dy"], 9, 1))
        out = torch.full((want.numel(),), 0.5, device="cuda")
        jobs.append(BlockWgradJob(t["x"].data_ptr(), t["h"].data_ptr(), t["g"].data_ptr(), t["dy"].data_ptr(), t["da"].data_ptr(),
                                  t["dz1"].data_ptr(), out.data_ptr(), C_, T, dil, Ci))
        keep.append(t); outs.append(out); wants.append(want + 0.5)
    arr = (BlockWgradJob * len(jobs))(*jobs)
    rc = lib.nsc_gated_block_wgrad_batch(arr, len(jobs), B, 20, 9, ws.data_ptr(), nws, _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for i, (w_, o_) in enumerate(zip(wants, outs)):
        assert_close(o_.cpu().numpy(), w_.cpu().numpy(), tol=2e-4, what=f"batched block wgrad job {i} {shapes[i]}")


def test_conv_wgrad_batch_equals_per_conv_launches(lib):
    """nsc_conv1d_wgrad_batch (deferred; mixed shapes and kernel classes, incl. the role-swapped Cout == 1 form, stride 2 and a
    bias-less job) == one nsc_conv1d_wgrad per conv."""
    from nsc_amd._lib import ConvWgradJob
    rng = np.random.default_rng(99)
    B = 5
    # (Cin, Cout, T, K, dil, stride, swapped)
    shapes = [(1, 100, 512, 55, 1, 1, 0), (100, 100, 256, 1, 1, 1, 0), (20, 100, 256, 9, 1, 1, 0), (20, 20, 256, 15, 2, 1, 0),
              (1, 20, 256, 1, 1, 1, 0), (100, 100, 512, 9, 1, 2, 0), (100, 1, 256, 55, 1, 1, 1), (50, 1, 512, 55, 1, 1, 1),
              (20, 20, 256, 15, 1, 1, 0), (1, 100, 256, 55, 1, 1, 0), (7, 13, 300, 5, 3, 1, 0), (100, 100, 256, 1, 1, 1, 0),
              (20, 100, 256, 9, 1, 1, 0)]
    jobs, keep, refs, outs = [], [], [], []
    for (Cin, Cout, T, K, dil, s, sw) in shapes:
        Tout, padL, _ = O.same_pad(T, K, dil, s)
        x = dev(rng.standard_normal((B, Cin, T)).astype(np.float32))
        dz = dev(rng.standard_normal((B, Cout, Tout)).astype(np.float32))
        dw_ref = torch.full((K, Cin, Cout), 0.25, device="cuda"); db_ref = torch.full((Cout,), 0.25, device="cuda")
        dw = dw_ref.clone(); db = db_ref.clone()
        if sw:   # Cout == 1: "input" = dz, "grad" = x, flipped taps, no bias row
            d = _desc(B=B, Cin=1, Cout=Cin, Tin=Tout, Tout=T, K=K, dil=dil, padL=(K - 1) * dil - padL)
            assert lib.nsc_conv1d_wgrad(C.byref(d), dz.data_ptr(), x.data_ptr(), dw_ref.data_ptr(), None, 1, _st()) == 0
            jobs.append(ConvWgradJob(d, dz.data_ptr(), x.data_ptr(), dw.data_ptr(), None, 1))
        else:
            d = _desc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=Tout, K=K, dil=dil, stride=s, padL=padL)
            assert lib.nsc_conv1d_wgrad(C.byref(d), x.data_ptr(), dz.data_ptr(), dw_ref.data_ptr(), db_ref.data_ptr(), 0, _st()) == 0
            jobs.append(ConvWgradJob(d, x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0))
        keep += [x, dz]; refs.append((dw_ref, db_ref)); outs.append((dw, db))
    arr = (ConvWgradJob * len(jobs))(*jobs)
    lib.nsc_conv1d_wgrad_batch_workspace.restype = C.c_long
    nws = lib.nsc_conv1d_wgrad_batch_workspace(arr, len(jobs))
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device="cuda")
    rc = lib.nsc_conv1d_wgrad_batch(arr, len(jobs), ws.data_ptr(), nws, _st())
    assert rc == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for i, ((rw, rb), (ow, ob)) in enumerate(zip(refs, outs)):
        assert_close(ow.cpu().numpy() - 0.25, rw.cpu().numpy() - 0.25, tol=2e-4, what=f"batched conv wgrad dW job {i} {shapes[i]}")
        assert_close(ob.cpu().numpy() - 0.25, rb.cpu().numpy() - 0.25, tol=2e-4, what=f"batched conv wgrad db job {i} {shapes[i]}")


@pytest.mark.parametrize("shape", [(7, 256, 32), (3, 16, 256), (4, 128, 32), (2, 50, 24)])
def test_frame_entropy(lib, shape):
    """nsc_frame_entropy == entropy_coding_loss (loss_terms_and_measures.py:262-267) applied to one frame at a time."""
    B, L, nb = shape
    rng = np.random.default_rng(B * 1000 + L + nb)
    z = rng.standard_normal((B, L, nb)) * 3
    p = np.exp(z - z.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    p[0] = 0; p[0, :, 3] = 1.0                      # a frame that uses one bin: entropy ~ 0
    ent = torch.full((B,), float("nan"), device="cuda")
    assert lib.nsc_frame_entropy(P(p.astype(np.float32)), B, L, nb, ent.data_ptr(), _st()) == 0
    ref = np.array([O.entropy_coding_loss(p[b:b + 1]) for b in range(B)])
    got = ent.cpu().numpy()
    assert np.all(np.abs(got - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))), (got, ref)


def test_stride2_dgrad_polyphase_equals_autograd(lib):
    """Data gradient of the stride-2 k9 conv as a 5-tap stride-1 conv over dy with sub-pixel-shuffled output (engine
    _Conv.wtpoly_index) vs autograd of the oracle conv; also exercises the interleaved row-wise shuffle epilogue with aux."""
    rng = np.random.default_rng(2024)
    B, Cin, Cout, T, K = 3, 100, 100, 512, 9
    x = rng.standard_normal((B, T, Cin)).astype(np.float32)
    W = (rng.standard_normal((K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    Tout, padL, _ = O.same_pad(T, K, 1, 2)
    assert (Tout, padL) == (256, 3)
    dz = rng.standard_normal((B, Tout, Cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    (OT.conv1d(xt, torch.tensor(W, dtype=torch.float64), torch.zeros(Cout, dtype=torch.float64), 1, 2, None)
     * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    Wp = np.zeros((5, Cout, 2 * Cin), np.float32)
    for tp in range(5):
        for par in range(2):
            k = 7 - 2 * tp + par
            if 0 <= k < K:
                Wp[tp, :, par::2] = W[k].T
    aux = rng.standard_normal((B, T, Cin)).astype(np.float32)          # laid out like the OUTPUT
    dx = torch.full((B, Cin, T), float("nan"), device="cuda")
    d = _desc(B=B, Cin=Cout, Cout=2 * Cin, Tin=Tout, Tout=Tout, K=5, dil=1, stride=1, padL=2, out_mode=1, mul_mode=1)
    rc = lib.nsc_conv1d_fwd(C.byref(d), P(dz.transpose(0, 2, 1)), P(Wp), None, None, P(aux.transpose(0, 2, 1)), dx.data_ptr(), _st())
    assert rc == 0, lib.nsc_last_error()
    ref = xt.grad.numpy() * np.where(aux > 0, 1.0, 0.2)
    assert_close(dx.cpu().numpy().transpose(0, 2, 1), ref, tol=2e-4, what="polyphase stride-2 dgrad")


def test_sum_all_batch_and_entropy_batch_equal_single_launches(lib):
    """nsc_sum_all_batch / nsc_entropy_from_hist_batch (one launch for a step's bias sums / quantizer entropies) give what
    the per-tensor entries give; the entropy values are checked against the oracle."""
    from nsc_amd._lib import EntropyJob, SumJob
    rng = np.random.default_rng(91)
    xs = [dev(rng.standard_normal(n).astype(np.float32)) for n in (65536, 300, 1, 32768, 70000)]
    one = [torch.full((1,), 0.25, device="cuda") for _ in xs]
    bat = [torch.full((1,), 0.25, device="cuda") for _ in xs]
    for x, o in zip(xs, one):
        assert lib.nsc_sum_all(x.data_ptr(), o.data_ptr(), x.numel(), _st()) == 0
    jobs = (SumJob * len(xs))(*[SumJob(x.data_ptr(), o.data_ptr(), x.numel()) for x, o in zip(xs, bat)])
    assert lib.nsc_sum_all_batch(jobs, len(xs), _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for x, a, b in zip(xs, one, bat):
        want = float(x.double().sum()) + 0.25
        assert abs(float(b) - want) <= 2e-5 * float(x.abs().sum()) + 1e-6 and abs(float(a) - float(b)) <= 2e-5 * float(x.abs().sum())
    hists = [dev((rng.random(nb) * 50 + 0.01).astype(np.float32)) for nb in (32, 256, 32, 7)]
    ent1, gh1, ent2, gh2 = [], [], [], []
    for h in hists:
        e, g = torch.empty(1, device="cuda"), torch.empty(h.numel(), device="cuda")
        assert lib.nsc_entropy_from_hist(h.data_ptr(), h.numel(), e.data_ptr(), g.data_ptr(), _st()) == 0
        ent1.append(e); gh1.append(g)
        ent2.append(torch.empty(1, device="cuda")); gh2.append(torch.empty(h.numel(), device="cuda"))
    ej = (EntropyJob * len(hists))(*[EntropyJob(h.data_ptr(), e.data_ptr(), g.data_ptr(), h.numel())
                                     for h, e, g in zip(hists, ent2, gh2)])
    assert lib.nsc_entropy_from_hist_batch(ej, len(hists), _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for h, e1, g1, e2, g2 in zip(hists, ent1, gh1, ent2, gh2):
        assert torch.equal(e1, e2) and torch.equal(g1, g2)
        hn = h.cpu().numpy().astype(np.float64)
        pr = hn / hn.sum()
        assert abs(float(e2) - float(-(pr * np.log2(pr + 1e-7)).sum())) < 1e-4


@pytest.mark.parametrize("shape", [(3, 100, 256), (2, 50, 128), (2, 100, 96), (1, 50, 520)])
def test_upsample_stage_fwd_bwd(lib, shape):
    """nsc_upsample_fwd / nsc_upsample_bwd (depthwise k9 -> pointwise -> leaky-relu -> sub-pixel shuffle in one kernel, and
    its data-path backward) vs float64 autograd of the same three ops; incl. a ragged last tile (T = 96, 520)."""
    import torch.nn.functional as F
    B, C_, T = shape
    rng = np.random.default_rng(500 + C_ + T)
    x = rng.standard_normal((B, C_, T)).astype(np.float32)
    wd = (0.3 * rng.standard_normal((9, C_))).astype(np.float32)            # depthwise_kernel [9, C, 1]
    wp = (0.1 * rng.standard_normal((C_, C_))).astype(np.float32)           # pointwise_kernel [1, Cin, Cout]
    bias = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    dz = rng.standard_normal((B, C_ // 2, 2 * T)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    dwo = F.conv1d(xt, torch.tensor(wd.T.copy(), dtype=torch.float64).reshape(C_, 1, 9), padding=4, groups=C_)
    dwo.retain_grad()
    pre = torch.einsum("bit,io->bot", dwo, torch.tensor(wp, dtype=torch.float64)) + torch.tensor(bias, dtype=torch.float64)[None, :, None]
    shuf = lambda z: z.reshape(B, C_ // 2, 2, T).permute(0, 1, 3, 2).reshape(B, C_ // 2, 2 * T)   # [b, oc, 2t + par] = z[b, 2oc + par, t]
    ypre = shuf(pre)
    (ypre * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    want_y = F.leaky_relu(ypre, 0.2).detach().numpy()
    xd, wdd, wpd, bd, dzd = dev(x), dev(wd), dev(wp), dev(bias), dev(dz)
    y = torch.full((B, C_ // 2, 2 * T), float("nan"), device="cuda")
    dw_out = torch.full((B, C_, T), float("nan"), device="cuda")
    rc = lib.nsc_upsample_fwd(xd.data_ptr(), wdd.data_ptr(), wpd.data_ptr(), bd.data_ptr(), dw_out.data_ptr(), y.data_ptr(),
                              B, C_, T, 9, 2, _st())
    assert rc == 0, lib.nsc_last_error()
    assert_close(dw_out.cpu().numpy(), dwo.detach().numpy(), what=f"upsample depthwise out {shape}")
    assert_close(y.cpu().numpy(), want_y, what=f"upsample y {shape}")
    y2 = torch.full_like(y, float("nan"))
    assert lib.nsc_upsample_fwd(xd.data_ptr(), wdd.data_ptr(), wpd.data_ptr(), bd.data_ptr(), None, y2.data_ptr(), B, C_, T, 9, 2,
                                _st()) == 0
    assert torch.equal(y2, y)                                                  # inference form: no saved depthwise output
    dzp, ddw, dx = (torch.full((B, C_, T), float("nan"), device="cuda") for _ in range(3))
    rc = lib.nsc_upsample_bwd(dzd.data_ptr(), wdd.data_ptr(), wpd.data_ptr(), dzp.data_ptr(), ddw.data_ptr(), dx.data_ptr(),
                              B, C_, T, 9, _st())
    assert rc == 0, lib.nsc_last_error()
    want_dzp = dz.reshape(B, C_ // 2, T, 2).transpose(0, 1, 3, 2).reshape(B, C_, T)
    assert np.array_equal(dzp.cpu().numpy(), want_dzp)
    assert_close(ddw.cpu().numpy(), dwo.grad.numpy(), what=f"upsample ddw {shape}")
    assert_close(dx.cpu().numpy(), xt.grad.numpy(), what=f"upsample dx {shape}")


@pytest.mark.parametrize("C_,Cin,dil,T,B", [(100, 100, 1, 256, 3), (100, 100, 2, 512, 2), (50, 50, 1, 512, 2), (50, 50, 2, 200, 3),
                                           (100, 1, 1, 256, 2), (100, 1, 2, 300, 3), (50, 1, 1, 130, 2), (50, 1, 2, 512, 70),
                                           (25, 25, 1, 128, 3), (25, 25, 2, 128, 300), (25, 1, 2, 128, 5),
                                           (100, 100, 2, 130, 3), (25, 25, 1, 67, 3)])
def test_block_kernels_on_parameter_images_equal_the_plain_entry_points(lib, C_, Cin, dil, T, B):
    """nsc_gated_block_fwd_img / _dgrad_img (fast prologue from a kernel-ready image built by nsc_gated_block_image_index +
    nsc_gather) produce the same bits as nsc_gated_block_fwd[_cin1] / nsc_gated_block_dgrad[_cin1] on the same parameters."""
    import ctypes as C
    rng = np.random.default_rng(C_ + Cin + dil + T)
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w1, b1, wl, bl, wr, br, w9, b9 = f(1, Cin, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)
    flat = np.concatenate([a.reshape(-1) for a in (w1, b1, wl, bl, wr, br, w9, b9)])
    sizes = [a.size for a in (w1, b1, wl, bl, wr, br, w9, b9)]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pd = dev(flat)
    pw = lambda i: pd.data_ptr() + 4 * int(offs[i])

    def image(which, src, offsets):
        n = int(lib.nsc_gated_block_image_floats(which, C_, Cin, dil))
        assert n > 0
        idx = np.empty(n, np.int32)
        assert lib.nsc_gated_block_image_index(which, C_, Cin, dil, (C.c_long * len(offsets))(*[int(o) for o in offsets]),
                                               idx.ctypes.data_as(C.c_void_p)) == 0, lib.nsc_last_error()
        img = torch.empty(n, device="cuda")
        idx_d = torch.tensor(idx, device="cuda")
        assert lib.nsc_gather(src.data_ptr(), idx_d.data_ptr(), img.data_ptr(), n, _st()) == 0
        return img

    x = dev(rng.standard_normal((B, Cin, T)).astype(np.float32))
    outs = {}
    for use_img in (False, True):
        out = torch.full((B, C_, T), float("nan"), device="cuda")
        sv = [torch.full((B, 20, T), float("nan"), device="cuda") for _ in range(4)]
        if use_img:
            img = image(0, pd, offs)
            rc = lib.nsc_gated_block_fwd_img(img.data_ptr(), x.data_ptr(), out.data_ptr(), *[t.data_ptr() for t in sv], B, C_, Cin, T, dil,
                                             0, _st())
        else:
            fn = lib.nsc_gated_block_fwd_cin1 if Cin == 1 else lib.nsc_gated_block_fwd
            rc = fn(x.data_ptr(), pw(0), pw(1), pw(2), pw(3), pw(4), pw(5), pw(6), pw(7), out.data_ptr(), *[t.data_ptr() for t in sv],
                    B, C_, T, 20, 9, dil, 0, _st())
        assert rc == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        outs[use_img] = [out] + sv
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    # ---- data gradient: flipped / transposed kernels, then the same comparison ----
    wt = {}
    for name, w in (("1", w1), ("l", wl), ("r", wr), ("9", w9)):
        wt[name] = np.ascontiguousarray(w[::-1].transpose(0, 2, 1))
    tflat = np.concatenate([wt[k].reshape(-1) for k in ("1", "l", "r", "9")])
    toffs = np.concatenate([[0], np.cumsum([wt[k].size for k in ("1", "l", "r", "9")])[:-1]]).astype(np.int64)
    td = dev(tflat)
    tw = lambda i: td.data_ptr() + 4 * int(toffs[i])
    h, lin = dev(rng.standard_normal((B, 20, T)).astype(np.float32)), dev(rng.standard_normal((B, 20, T)).astype(np.float32))
    th = torch.tanh(dev(rng.standard_normal((B, 20, T)).astype(np.float32)))
    dy = dev(rng.standard_normal((B, C_, T)).astype(np.float32))
    res = {}
    for use_img in (False, True):
        dx = torch.full((B, Cin, T), float("nan"), device="cuda")
        da = torch.full((B, 40, T), float("nan"), device="cuda")
        dz1 = torch.full((B, 20, T), float("nan"), device="cuda")
        dl, dg = da.data_ptr(), da.data_ptr() + 4 * 20 * T
        if use_img:
            img = image(1, td, toffs)
            rc = lib.nsc_gated_block_dgrad_img(img.data_ptr(), None if Cin == 1 else x.data_ptr(), h.data_ptr(), lin.data_ptr(),
                                               th.data_ptr(), dy.data_ptr(), dx.data_ptr(), dl, dg, dz1.data_ptr(), B, C_, Cin, T, dil,
                                               0 if Cin == 1 else 2, 40, _st())
        elif Cin == 1:
            rc = lib.nsc_gated_block_dgrad_cin1(h.data_ptr(), lin.data_ptr(), th.data_ptr(), dy.data_ptr(), tw(0), tw(1), tw(2), tw(3),
                                                dx.data_ptr(), dl, dg, dz1.data_ptr(), B, C_, T, 20, 9, dil, 40, _st())
        else:
            rc = lib.nsc_gated_block_dgrad(x.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), dy.data_ptr(), tw(0), tw(1), tw(2),
                                           tw(3), dx.data_ptr(), da.data_ptr(), dz1.data_ptr(), B, C_, T, 20, 9, dil, 2, _st())
        assert rc == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        res[use_img] = (dx, da, dz1)
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())


@pytest.mark.parametrize("C_,T,B,Cin0", [(100, 256, 3, 100), (100, 512, 2, 100), (50, 512, 3, 50), (50, 200, 3, 50), (25, 128, 5, 25),
                                         (100, 132, 3, 100), (25, 68, 3, 25), (100, 256, 128, 100), (50, 512, 128, 50), (100, 64, 300, 100),
                                         (100, 256, 3, 1), (100, 256, 128, 1), (50, 512, 5, 1), (25, 128, 70, 1)])
def test_pair_launches_equal_two_single_launches(lib, C_, T, B, Cin0):
    """nsc_gated_block_pair_fwd_img / _dgrad_img (the dil-1 and dil-2 block of a stack in ONE launch, neighbour flags between the
    workgroups instead of a kernel boundary) produce the same bits as the two blocks launched one after the other - at sizes
    with one tile per workgroup, with chains inside a frame, with more tiles than workgroups, and at the headline batch; no
    neighbour wait may time out (the caller-owned counter stays 0).  Run three times: the result must not depend on workgroup timing.
    (T % 4 == 0: other lengths are refused with NSC_ERR_UNSUPPORTED and the engine launches the blocks one by one.)"""
    import ctypes as C
    rng = np.random.default_rng(C_ + T + B + Cin0)
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    nfl = int(lib.nsc_gated_block_pair_flag_ints())
    tmo = torch.zeros(4, dtype=torch.int32, device="cuda")       # the sticky time-out counter (the library only adds to it)
    blocks = []
    for dil in (1, 2):
        Ci = Cin0 if dil == 1 else C_                 # (Cin0 = 1: the first block of a decoder stage)
        w = [f(1, Ci, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
        flat = np.concatenate([a.reshape(-1) for a in w])
        offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
        wt = [np.ascontiguousarray(w[i][::-1].transpose(0, 2, 1)) for i in (0, 2, 4, 6)]
        tflat = np.concatenate([a.reshape(-1) for a in wt])
        toffs = np.concatenate([[0], np.cumsum([a.size for a in wt])[:-1]]).astype(np.int64)
        imgs = []
        for which, src, o in ((0, dev(flat), offs), (1, dev(tflat), toffs)):
            n = int(lib.nsc_gated_block_image_floats(which, C_, Ci, dil))
            idx = np.empty(n, np.int32)
            assert lib.nsc_gated_block_image_index(which, C_, Ci, dil, (C.c_long * len(o))(*[int(v) for v in o]),
                                                   idx.ctypes.data_as(C.c_void_p)) == 0, lib.nsc_last_error()
            img = torch.empty(n, device="cuda")
            assert lib.nsc_gather(src.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, _st()) == 0
            imgs.append(img)
        blocks.append(imgs)
    (f0, b0), (f1, b1) = blocks
    x = dev(rng.standard_normal((B, Cin0, T)).astype(np.float32))
    nan = lambda *sh: torch.full(sh, float("nan"), device="cuda")
    P = lambda t: t.data_ptr()
    # ---- forward ----
    o0, o1 = nan(B, C_, T), nan(B, C_, T)
    s0, s1 = [nan(B, 20, T) for _ in range(4)], [nan(B, 20, T) for _ in range(4)]
    assert lib.nsc_gated_block_fwd_img(P(f0), P(x), P(o0), *[P(t) for t in s0], B, C_, Cin0, T, 1, 0, _st()) == 0, lib.nsc_last_error()
    assert lib.nsc_gated_block_fwd_img(P(f1), P(o0), P(o1), *[P(t) for t in s1], B, C_, C_, T, 2, 1, _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for rep in range(3):
        p0, p1 = nan(B, C_, T), nan(B, C_, T)
        q0, q1 = [nan(B, 20, T) for _ in range(4)], [nan(B, 20, T) for _ in range(4)]
        flags = torch.zeros(nfl, dtype=torch.int32, device="cuda")
        assert lib.nsc_gated_block_pair_fwd_img(P(f0), P(f1), P(x), P(p0), *[P(t) for t in q0], P(p1), *[P(t) for t in q1], B, C_, Cin0, T, 1,
                                                P(flags), P(tmo), _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        assert int(tmo[0]) == 0, "a neighbour wait timed out"
        for a, b in zip([o0, o1] + s0 + s1, [p0, p1] + q0 + q1):
            assert torch.equal(a, b) and bool(torch.isfinite(a).all()), ("forward", rep)
    # ---- data gradient: block 1 (dil 2) first, block 0 on its dx ----
    dy = dev(rng.standard_normal((B, C_, T)).astype(np.float32))
    h0, l0, h1, l1 = (dev(rng.standard_normal((B, 20, T)).astype(np.float32)) for _ in range(4))
    t0_, t1_ = (torch.tanh(dev(rng.standard_normal((B, 20, T)).astype(np.float32))) for _ in range(2))
    x1 = dev(rng.standard_normal((B, C_, T)).astype(np.float32))
    dx1, da1, dz1 = nan(B, C_, T), nan(B, 40, T), nan(B, 20, T)
    dx0, da0, dz0 = nan(B, Cin0, T), nan(B, 40, T), nan(B, 20, T)
    act0 = 0 if Cin0 == 1 else 2
    assert lib.nsc_gated_block_dgrad_img(P(b1), P(x1), P(h1), P(l1), P(t1_), P(dy), P(dx1), P(da1), P(da1) + 80 * T, P(dz1), B, C_, C_, T, 2,
                                         2, 40, _st()) == 0, lib.nsc_last_error()
    assert lib.nsc_gated_block_dgrad_img(P(b0), None if Cin0 == 1 else P(x), P(h0), P(l0), P(t0_), P(dx1), P(dx0), P(da0), P(da0) + 80 * T,
                                         P(dz0), B, C_, Cin0, T, 1, act0, 40, _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for rep in range(3):
        e1, a1, z1 = nan(B, C_, T), nan(B, 40, T), nan(B, 20, T)
        e0, a0, z0 = nan(B, Cin0, T), nan(B, 40, T), nan(B, 20, T)
        flags = torch.zeros(nfl, dtype=torch.int32, device="cuda")
        assert lib.nsc_gated_block_pair_dgrad_img(P(b1), P(x1), P(h1), P(l1), P(t1_), P(dy), P(e1), P(a1), P(z1), P(b0),
                                                  None if Cin0 == 1 else P(x), P(h0), P(l0), P(t0_), P(e0), P(a0), P(z0), B, C_, Cin0, T, act0,
                                                  P(flags), P(tmo), _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        assert int(tmo[0]) == 0, "a neighbour wait timed out"
        for a, b in zip([dx1, da1, dz1, dx0, da0, dz0], [e1, a1, z1, e0, a0, z0]):
            assert torch.equal(a, b) and bool(torch.isfinite(a).all()), ("dgrad", rep)


@pytest.mark.parametrize("B", [2, 128])
def test_conv_wgrad_batch_with_one_input_channel_convs(lib, B):
    """nsc_conv1d_wgrad_batch over the headline step's job mix - two k55 1 -> 100 input convs (their own swapped-role kernel:
    rows = output channels, columns = taps, the bias as column K), a k55 1 -> 50 conv WITHOUT a bias, a T not a multiple of 256
    (stays on the generic kernel), a pointwise 100 -> 100 conv and a k55 100 -> 1 conv - against float64 on the host; gradients
    ACCUMULATE into dw / db (started from a non-zero value); taps flipped for one job."""
    import ctypes as C
    from nsc_amd._lib import ConvWgradJob
    rng = np.random.default_rng(B)
    # (Cin, Cout, T, K, padL, bias, flip)
    specs = [(1, 100, 512, 55, 27, True, 0), (1, 100, 512, 55, 27, True, 1), (1, 50, 256, 55, 27, False, 0), (1, 100, 320, 55, 27, True, 0),
             (100, 100, 256, 1, 0, True, 0), (100, 1, 256, 55, 27, True, 0)]
    jobs, refs, outs = [], [], []
    for (Cin, Cout, T, K, padL, bias, flip) in specs:
        x = rng.standard_normal((B, Cin, T)).astype(np.float32)
        dz = rng.standard_normal((B, Cout, T)).astype(np.float32)
        dw0 = rng.standard_normal((K, Cin, Cout)).astype(np.float32)
        db0 = rng.standard_normal((Cout,)).astype(np.float32)
        xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (padL, K)))
        dwr = np.stack([np.einsum("bit,bot->io", xp[:, :, k:k + T], dz.astype(np.float64)) for k in range(K)], 0)
        if flip:
            dwr = dwr[::-1]
        refs.append((dw0 + dwr, db0 + dz.astype(np.float64).sum((0, 2))))
        xd, dzd, dwd, dbd = dev(x), dev(dz), dev(dw0), dev(db0)
        outs.append((dwd, dbd, bias))
        d = _desc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=T, K=K, padL=padL)
        jobs.append((ConvWgradJob(d, xd.data_ptr(), dzd.data_ptr(), dwd.data_ptr(), dbd.data_ptr() if bias else None, flip), xd, dzd))
    arr = (ConvWgradJob * len(jobs))(*[j[0] for j in jobs])
    lib.nsc_conv1d_wgrad_batch_workspace.restype = C.c_long
    need = int(lib.nsc_conv1d_wgrad_batch_workspace(arr, len(jobs)))
    ws = torch.full((need,), float("nan"), device="cuda")
    assert lib.nsc_conv1d_wgrad_batch(arr, len(jobs), ws.data_ptr(), need, _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for (dwd, dbd, bias), (dwr, dbr), sp in zip(outs, refs, specs):
        assert_close(dwd.cpu().numpy(), dwr, tol=2e-5, what=f"dW of {sp}")
        if bias:
            assert_close(dbd.cpu().numpy(), dbr, tol=2e-5, what=f"db of {sp}")


@pytest.mark.parametrize("C_,T,B,soft", [(100, 256, 128, True), (100, 256, 3, False), (50, 128, 5, True), (100, 200, 3, True)])
def test_cout1_conv_with_fused_quantizer_equals_conv_then_quantizer(lib, C_, T, B, soft):
    """nsc_conv1d_cout1_fwd_quant (the encoder's k55 C -> 1 conv + tanh with the training-shape quantizer in the same launch)
    against nsc_conv1d_cout1_fwd followed by nsc_quantize_fwd: the float codes bit for bit; the quantised codes, quan_loss and the
    soft histogram up to the summation order of a 4-lane instead of an 8-lane group (hard codes: exactly the same bin); quan
    accumulates into a zeroed buffer, the histogram onto what is there."""
    import ctypes as C
    from nsc_amd._lib import Cout1Quant
    rng = np.random.default_rng(C_ + T + B)
    x = dev(rng.standard_normal((B, C_, T)).astype(np.float32))
    w = dev((0.02 * rng.standard_normal((55, C_, 1))).astype(np.float32))
    bias = dev(np.array([0.01], np.float32))
    alpha, bins = dev(np.array([-300.0 if soft else -20.0], np.float32)), dev(np.linspace(-1, 1, 32).astype(np.float32))
    d = _desc(B=B, Cin=C_, Cout=1, Tin=T, Tout=T, K=55, padL=27, act=1)
    code0, code1 = torch.full((B, 1, T), float("nan"), device="cuda"), torch.full((B, 1, T), float("nan"), device="cuda")
    q0, q1 = torch.full((B, 1, T), float("nan"), device="cuda"), torch.full((B, 1, T), float("nan"), device="cuda")
    quan0, quan1 = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    h0, h1 = torch.ones(32, device="cuda"), torch.ones(32, device="cuda")
    assert lib.nsc_conv1d_cout1_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, None, code0.data_ptr(), _st()) == 0
    assert lib.nsc_quantize_fwd(code0.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, int(soft), B, T, 32, None, q0.data_ptr(),
                                quan0.data_ptr(), h0.data_ptr(), _st()) == 0, lib.nsc_last_error()
    qz = Cout1Quant(alpha.data_ptr(), bins.data_ptr(), 1.0, int(soft), 32, q1.data_ptr(), quan1.data_ptr(), h1.data_ptr())
    assert lib.nsc_conv1d_cout1_fwd_quant(C.byref(d), x.data_ptr(), w.data_ptr(), bias.data_ptr(), code1.data_ptr(), C.byref(qz),
                                          _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    assert torch.equal(code0, code1) and bool(torch.isfinite(code1).all())
    if soft:
        assert_close(q1.cpu().numpy(), q0.cpu().numpy(), tol=2e-6, atol=1e-7, what="soft codes")
    else:
        assert torch.equal(q0, q1), "hard codes"
    assert_close(quan1.cpu().numpy(), quan0.cpu().numpy(), tol=2e-6, what="quan_loss per frame")
    assert_close(h1.cpu().numpy(), h0.cpu().numpy(), tol=1e-5, what="soft histogram")
