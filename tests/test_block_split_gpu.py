"""GPU parity tests of the SPLIT-operand gated-block kernels (csrc/block_split.hip: bf16 matrix cores, every fp32 operand split into
three bf16 pieces, six products, fp32 accumulation) against the float64 oracle - at the SAME bounds as the exact-fp32 kernels - and
against the exact kernels themselves (the achieved error of both arms against float64 is reported; the split arm must stay within
1.5x of the exact arm + a floor of one fp32 rounding of the tensor's scale)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from tests._util import assert_close, dev
from tests.test_kernels_gpu import P, _st, lib  # noqa: F401  (the NaN-poisoning library proxy)

pytestmark = pytest.mark.gpu


def _params(rng, Cin, C_):
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, Cin, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    flat = np.concatenate([a.reshape(-1) for a in w])
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    return w, flat, offs


def _image(lib, split, which, C_, Cin, dil, src, offs):
    fn_n = lib.nsc_gated_block_simage_words if split else lib.nsc_gated_block_image_floats
    fn_i = lib.nsc_gated_block_simage_index if split else lib.nsc_gated_block_image_index
    n = int(fn_n(which, C_, Cin, dil))
    assert n > 0
    idx = np.empty(n, np.int32)
    assert fn_i(which, C_, Cin, dil, (C.c_long * len(offs))(*[int(o) for o in offs]), idx.ctypes.data_as(C.c_void_p)) == 0, lib.nsc_last_error()
    img = torch.empty(n, device="cuda")
    assert lib.nsc_gather(src.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, _st()) == 0
    return img


def _ref_block(x, w, dil, flat):
    """float64 gated_bottleneck (nn_core_operator.py:82-112) on [B, C, T] tensors; returns h, lin, th, g, out."""
    w1, b1, wl, bl, wr, br, w9, b9 = [a.astype(np.float64) for a in w]
    B, Cin, T = x.shape
    xd = x.astype(np.float64)
    h = np.einsum("bit,io->bot", xd, w1[0]) + b1[None, :, None]
    h = np.where(h > 0, h, 0.2 * h)

    def conv(v, k, d):
        K = k.shape[0]
        pad = (K - 1) * d
        pl = pad // 2
        vp = np.pad(v, ((0, 0), (0, 0), (pl, pad - pl)))
        return sum(np.einsum("bit,io->bot", vp[:, :, q * d:q * d + T], k[q]) for q in range(K))
    lin = conv(h, wl, dil) + bl[None, :, None]
    th = np.tanh(conv(h, wr, dil) + br[None, :, None])
    g = lin * th
    y = conv(g, w9, 1) + b9[None, :, None] + xd                      # (Cin = 1: broadcast residual)
    out = y if flat else np.where(y > 0, y, 0.2 * y)
    return h, lin, th, g, out


_CASES = [(2, 100, 100, 512, 2, 0), (2, 100, 100, 256, 1, 1), (3, 50, 50, 512, 2, 0), (2, 50, 50, 512, 1, 1), (1, 100, 100, 200, 2, 0),
          (70, 100, 100, 300, 2, 0), (40, 50, 50, 512, 1, 1),
          # long chains of consecutive tiles (carried h / g columns), crossing frame boundaries
          (150, 100, 100, 512, 1, 0), (150, 50, 50, 512, 2, 1), (72, 100, 100, 256, 2, 0),
          # C = 25 (third resolution of '2 2' codecs)
          (3, 25, 25, 128, 1, 0), (300, 25, 25, 128, 2, 1), (2, 25, 25, 200, 2, 0),
          # short frames, T a multiple of 4 but not of the 64-step tile
          (3, 100, 100, 132, 2, 0), (40, 100, 100, 204, 1, 1), (5, 50, 50, 68, 2, 0), (3, 25, 25, 36, 1, 0),
          # one input channel (first block of a decoder stage)
          (2, 100, 1, 256, 1, 0), (3, 100, 1, 300, 2, 0), (2, 50, 1, 132, 1, 0), (70, 50, 1, 512, 2, 0), (5, 25, 1, 128, 2, 1)]


@pytest.mark.parametrize("case", _CASES)
def test_split_gated_block_fwd_matches_the_float64_oracle(lib, case):
    B, C_, Cin, T, dil, flat = case
    rng = np.random.default_rng(100 + C_ + T + dil + Cin)
    w, pflat, offs = _params(rng, Cin, C_)
    pd = dev(pflat)
    x = rng.standard_normal((B, Cin, T)).astype(np.float32)
    xd = dev(x)
    ref = _ref_block(x, w, dil, flat)
    res = {}
    for split in (False, True):
        img = _image(lib, split, 0, C_, Cin, dil, pd, offs)
        out = torch.full((B, C_, T), float("nan"), device="cuda")
        sv = [torch.full((B, 20, T), float("nan"), device="cuda") for _ in range(4)]
        fn = lib.nsc_gated_block_fwd_simg if split else lib.nsc_gated_block_fwd_img
        assert fn(img.data_ptr(), xd.data_ptr(), out.data_ptr(), *[t.data_ptr() for t in sv], B, C_, Cin, T, dil, flat, _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        res[split] = [t.cpu().numpy() for t in sv + [out]]
        # without the optional outputs: same main output
        out2 = torch.full((B, C_, T), float("nan"), device="cuda")
        assert fn(img.data_ptr(), xd.data_ptr(), out2.data_ptr(), None, None, None, None, B, C_, Cin, T, dil, flat, _st()) == 0
        assert torch.equal(out, out2)
    names = ["h", "lin", "tanh", "g", "out"]
    for nm, a, r in zip(names, res[True], ref):
        assert_close(a, r, what=f"split block {nm} {case}")
    # error of both arms against float64, relative to the tensor's rms
    # (rms error within 1.5x of the exact arm; the maximum is one draw of a noisy statistic on the small tensors: 3x)
    for nm, e_, s_, r in zip(names, res[False], res[True], ref):
        rms = float(np.sqrt(np.mean(r ** 2)))
        ee, es = float(np.abs(e_ - r).max()) / rms, float(np.abs(s_ - r).max()) / rms
        re_, rs = float(np.sqrt(np.mean((e_ - r) ** 2))) / rms, float(np.sqrt(np.mean((s_ - r) ** 2))) / rms
        print(f"  {nm}: err / rms of the tensor   exact max {ee:.2e} rms {re_:.2e}   split max {es:.2e} rms {rs:.2e}")
        assert rs <= 1.5 * re_ + 1e-7 and es <= 3.0 * ee + 1e-6, (nm, ee, es, re_, rs)


def test_split_forward_refuses_rows_that_are_not_16_byte_aligned(lib):
    """The split kernels stage x by LDS-DMA in aligned 16-byte pieces: T % 4 != 0 is NSC_ERR_UNSUPPORTED (the engine keeps such shapes
    on the exact kernels), not a wrong result."""
    rng = np.random.default_rng(3)
    w, pflat, offs = _params(rng, 50, 50)
    img = _image(lib, True, 0, 50, 50, 1, dev(pflat), offs)
    x = dev(rng.standard_normal((2, 50, 70)).astype(np.float32))
    out = torch.zeros(2, 50, 70, device="cuda")
    assert lib.nsc_gated_block_fwd_simg(img.data_ptr(), x.data_ptr(), out.data_ptr(), None, None, None, None, 2, 50, 50, 70, 1, 0, _st()) == -2
    assert b"T % 4" in lib.nsc_last_error()


@pytest.mark.parametrize("C_,T,B,Cin0", [(100, 512, 2, 100), (100, 256, 5, 100), (50, 512, 3, 50), (25, 128, 9, 25), (100, 256, 128, 100),
                                         (50, 512, 128, 50), (100, 256, 3, 1), (100, 256, 128, 1), (50, 512, 5, 1), (25, 128, 70, 1)])
def test_split_pair_launch_equals_two_single_launches(lib, C_, T, B, Cin0):
    """nsc_gated_block_pair_fwd_simg (dilation 1 then 2 in ONE launch, neighbour flags) == the two split launches, bit for bit."""
    rng = np.random.default_rng(C_ + T + B + Cin0)
    nfl = int(lib.nsc_gated_block_pair_flag_ints())
    tmo = torch.zeros(4, dtype=torch.int32, device="cuda")
    imgs = []
    for dil in (1, 2):
        Ci = Cin0 if dil == 1 else C_
        w, pflat, offs = _params(rng, Ci, C_)
        imgs.append(_image(lib, True, 0, C_, Ci, dil, dev(pflat), offs))
    f0, f1 = imgs
    x = dev(rng.standard_normal((B, Cin0, T)).astype(np.float32))
    nan = lambda *sh: torch.full(sh, float("nan"), device="cuda")
    Pt = lambda t: t.data_ptr()
    o0, o1 = nan(B, C_, T), nan(B, C_, T)
    s0, s1 = [nan(B, 20, T) for _ in range(4)], [nan(B, 20, T) for _ in range(4)]
    assert lib.nsc_gated_block_fwd_simg(Pt(f0), Pt(x), Pt(o0), *[Pt(t) for t in s0], B, C_, Cin0, T, 1, 0, _st()) == 0, lib.nsc_last_error()
    assert lib.nsc_gated_block_fwd_simg(Pt(f1), Pt(o0), Pt(o1), *[Pt(t) for t in s1], B, C_, C_, T, 2, 1, _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for rep in range(3):
        p0, p1 = nan(B, C_, T), nan(B, C_, T)
        q0, q1 = [nan(B, 20, T) for _ in range(4)], [nan(B, 20, T) for _ in range(4)]
        flags = torch.zeros(nfl, dtype=torch.int32, device="cuda")
        assert lib.nsc_gated_block_pair_fwd_simg(Pt(f0), Pt(f1), Pt(x), Pt(p0), *[Pt(t) for t in q0], Pt(p1), *[Pt(t) for t in q1], B, C_, Cin0,
                                                 T, 1, Pt(flags), Pt(tmo), _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        assert int(tmo[0]) == 0, "a neighbour wait timed out"
        for a, b in zip([o0, o1] + s0 + s1, [p0, p1] + q0 + q1):
            assert torch.equal(a, b) and bool(torch.isfinite(a).all()), ("forward", rep)


def _wgrad_ref(t, C_, Cx, T, dil):
    """float64 parameter gradients of one gated block from its saved activations and data-path gradients (block.hip header)."""
    x, h, g, dy, da, dz1 = [t[k].astype(np.float64) for k in ("x", "h", "g", "dy", "da", "dz1")]
    gp_ = np.pad(g, ((0, 0), (0, 0), (4, 4)))
    hp_ = np.pad(h, ((0, 0), (0, 0), (7 * dil, 7 * dil)))
    dw9 = np.stack([np.einsum("bit,bot->io", gp_[:, :, k:k + T], dy) for k in range(9)], 0)
    dwl = np.stack([np.einsum("bit,bot->io", hp_[:, :, k * dil:k * dil + T], da[:, :20]) for k in range(15)], 0)
    dwr = np.stack([np.einsum("bit,bot->io", hp_[:, :, k * dil:k * dil + T], da[:, 20:]) for k in range(15)], 0)
    dw1 = np.einsum("bit,bot->io", x, dz1)
    parts = [dw1, dz1.sum((0, 2)), dwl, da[:, :20].sum((0, 2)), dwr, da[:, 20:].sum((0, 2)), dw9, dy.sum((0, 2))]
    return np.concatenate([p.reshape(-1) for p in parts])


@pytest.mark.parametrize("B,shapes", [(6, [(100, 100, 256, 2), (100, 100, 512, 1), (50, 50, 512, 2), (100, 1, 256, 1), (50, 50, 512, 1), (25, 25, 128, 2)]),
                                      (40, [(100, 100, 256, 1), (50, 1, 512, 2), (25, 25, 128, 1), (100, 100, 512, 2)]),
                                      (3, [(100, 100, 64, 2), (50, 50, 68, 1), (25, 1, 128, 2)])])
def test_split_block_wgrad_batch_matches_float64(lib, B, shapes):
    """nsc_gated_block_wgrad_batch_split (bf16 matrix cores, split operands) against float64 on the host and against the exact kernel:
    same bound as the exact kernel's test (2e-4 of each tensor's scale), rms error within 1.5x of the exact arm."""
    from nsc_amd._lib import BlockWgradJob
    rng = np.random.default_rng(B + len(shapes))
    lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
    nws = 2 * lib.nsc_gated_block_wgrad_batch_workspace(100)
    keep, refs = [], []
    outs = {False: [], True: []}
    jobs = {False: [], True: []}
    for (C_, Cx, T, dil) in shapes:
        th = {k: rng.standard_normal(s_).astype(np.float32) for k, s_ in
              dict(x=(B, Cx, T), h=(B, 20, T), g=(B, 20, T), dy=(B, C_, T), da=(B, 40, T), dz1=(B, 20, T)).items()}
        t = {k: dev(v) for k, v in th.items()}
        keep.append(t)
        refs.append(_wgrad_ref(th, C_, Cx, T, dil))
        n = refs[-1].size
        for split in (False, True):
            out = torch.full((n,), 0.5, device="cuda")
            outs[split].append(out)
            jobs[split].append(BlockWgradJob(t["x"].data_ptr(), t["h"].data_ptr(), t["g"].data_ptr(), t["dy"].data_ptr(), t["da"].data_ptr(),
                                             t["dz1"].data_ptr(), out.data_ptr(), C_, T, dil, Cx if Cx != C_ else 0))
    for split in (False, True):
        ws = torch.full((nws,), float("nan"), device="cuda")
        arr = (BlockWgradJob * len(shapes))(*jobs[split])
        fn = lib.nsc_gated_block_wgrad_batch_split if split else lib.nsc_gated_block_wgrad_batch
        assert fn(arr, len(shapes), B, 20, 9, ws.data_ptr(), nws, _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
    for i, sh in enumerate(shapes):
        C_, Cx, T, dil = sh
        r = refs[i]
        e_ = outs[False][i].cpu().numpy().astype(np.float64) - 0.5
        s_ = outs[True][i].cpu().numpy().astype(np.float64) - 0.5
        sizes = [Cx * 20, 20, 6000, 20, 6000, 20, 180 * C_, C_]
        o = 0
        for nm, n in zip(["dw1", "db1", "dwl", "dbl", "dwr", "dbr", "dw9", "db9"], sizes):
            rr, ee, ss = r[o:o + n], e_[o:o + n], s_[o:o + n]
            o += n
            assert_close(ss, rr, tol=2e-4, what=f"split wgrad {nm} job {i} {sh}")
            rms = float(np.sqrt(np.mean(rr ** 2)))
            re_, rs = float(np.sqrt(np.mean((ee - rr) ** 2))) / rms, float(np.sqrt(np.mean((ss - rr) ** 2))) / rms
            print(f"  job {i} {sh} {nm}: rms err / rms  exact {re_:.2e}  split {rs:.2e}")
            # (the 0.5 the gradients accumulate onto costs both arms the same fp32 rounding of the sum: floor 2e-7 of the tensor's rms)
            assert rs <= 1.5 * re_ + 2e-7 * max(1.0, 0.5 / rms), (nm, re_, rs)


def _dgrad_ref(w, x, h, lin, th, dy, dil, in_act):
    """float64 data-path backward of the gated block (block.hip header of the data-gradient kernels): dx, da (dlin | dgate), dz1."""
    w1, b1, wl, bl, wr, br, w9, b9 = [a.astype(np.float64) for a in w]
    x, h, lin, th, dy = [a.astype(np.float64) for a in (x, h, lin, th, dy)]
    B, C_, T = dy.shape

    def convT(v, k, d):          # gradient of `conv` (SAME, dilation d) with respect to its input
        K = k.shape[0]
        pad = (K - 1) * d
        pl = pad // 2
        out = np.zeros((B, k.shape[1], T + pad))
        for q in range(K):
            out[:, :, q * d:q * d + T] += np.einsum("bot,io->bit", v, k[q])
        return out[:, :, pl:pl + T]
    dg = convT(dy, w9, 1)
    dlin = dg * th
    dgate = dg * lin * (1.0 - th * th)
    dh = convT(dlin, wl, dil) + convT(dgate, wr, dil)
    dz1 = dh * np.where(h > 0, 1.0, 0.2)
    dx = np.einsum("bot,io->bit", dz1, w1[0]) + dy
    if in_act:
        dx = dx * np.where(x > 0, 1.0, 0.2)
    return dx, np.concatenate([dlin, dgate], 1), dz1


@pytest.mark.parametrize("case", [(2, 100, 100, 512, 2, 2), (2, 100, 100, 256, 1, 0), (3, 50, 50, 512, 2, 2), (2, 50, 50, 512, 1, 2),
                                  (1, 100, 100, 200, 2, 2), (70, 100, 100, 300, 2, 2), (40, 50, 50, 512, 1, 0), (150, 100, 100, 512, 1, 2),
                                  (150, 50, 50, 512, 2, 2), (72, 100, 100, 256, 2, 2), (3, 100, 100, 132, 2, 2), (40, 100, 100, 204, 1, 2),
                                  (5, 50, 50, 68, 2, 0), (128, 100, 100, 256, 2, 2), (3, 100, 1, 256, 1, 0), (130, 100, 1, 256, 1, 0),
                                  (2, 50, 1, 512, 1, 0), (128, 50, 50, 512, 1, 2), (128, 100, 100, 512, 2, 2)])
def test_three_launch_split_dgrad_matches_the_float64_oracle(lib, case):
    """nsc_gated_block_dgrad_simg2 (csrc/block_bwd_split.hip: the two long contractions as 80-row polyphase GEMMs on the bf16 matrix
    cores, then the 1x1 gradient + residual) against float64 and against the exact fused kernel on its image: the exact kernel's
    bounds; rms error within 1.5x of the exact arm.  Cin = 1: the first block of a decoder stage (broadcast residual)."""
    B, C_, Cin, T, dil, act = case
    rng = np.random.default_rng(11 + C_ + T + dil + B + Cin)
    w, pflat, offs = _params(rng, Cin, C_)
    r = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    x, h, lin, dy = r(B, Cin, T), r(B, 20, T), r(B, 20, T), r(B, C_, T)
    th = np.tanh(r(B, 20, T)).astype(np.float32)
    dxr, dar, dzr = _dgrad_ref(w, x if Cin > 1 else np.zeros((B, C_, T), np.float32), h, lin, th, dy, dil, act == 2)
    if Cin == 1:      # broadcast residual: dx = W1^T dz1 + sum_c dy
        dxr = np.einsum("bot,io->bit", dzr, w[0][0].astype(np.float64)) + dy.astype(np.float64).sum(1, keepdims=True)
    ref = (dxr, dar, dzr)
    xd, hd, ld, td, dyd = [dev(v) for v in (x, h, lin, th, dy)]
    pd = dev(pflat)
    wt = [np.ascontiguousarray(w[i][::-1].transpose(0, 2, 1)) for i in (0, 2, 4, 6)]
    tflat = np.concatenate([a.reshape(-1) for a in wt])
    toffs = np.concatenate([[0], np.cumsum([a.size for a in wt])[:-1]]).astype(np.int64)
    res = {}
    for split in (False, True):
        dx = torch.full((B, Cin, T), float("nan"), device="cuda")
        da = torch.full((B, 40, T), float("nan"), device="cuda")
        dz1 = torch.full((B, 20, T), float("nan"), device="cuda")
        if split:
            img = _image(lib, True, 2, C_, Cin, dil, pd, offs)
            assert lib.nsc_gated_block_dgrad_simg2(img.data_ptr(), pd.data_ptr() + 4 * int(offs[0]), xd.data_ptr() if Cin > 1 else None,
                                                   hd.data_ptr(), ld.data_ptr(), td.data_ptr(), dyd.data_ptr(), dx.data_ptr(), da.data_ptr(),
                                                   dz1.data_ptr(), B, C_, Cin, T, dil, act, _st()) == 0, lib.nsc_last_error()
        else:
            img = _image(lib, False, 1, C_, Cin, dil, dev(tflat), toffs)
            assert lib.nsc_gated_block_dgrad_img(img.data_ptr(), xd.data_ptr() if Cin > 1 else None, hd.data_ptr(), ld.data_ptr(),
                                                 td.data_ptr(), dyd.data_ptr(), dx.data_ptr(), da.data_ptr(), da.data_ptr() + 4 * 20 * T,
                                                 dz1.data_ptr(), B, C_, Cin, T, dil, act, 40, _st()) == 0, lib.nsc_last_error()
        torch.cuda.synchronize()
        res[split] = [t.cpu().numpy() for t in (dx, da, dz1)]
    for nm, a, r_ in zip(["dx", "da", "dz1"], res[True], ref):
        assert_close(a, r_, what=f"three-launch split dgrad {nm} {case}")
    for nm, e_, s_, r_ in zip(["dx", "da", "dz1"], res[False], res[True], ref):
        rms = float(np.sqrt(np.mean(r_ ** 2)))
        re_, rs = float(np.sqrt(np.mean((e_ - r_) ** 2))) / rms, float(np.sqrt(np.mean((s_ - r_) ** 2))) / rms
        print(f"  {nm}: rms err / rms  exact {re_:.2e}  split {rs:.2e}")
        assert rs <= 1.5 * re_ + 1e-7, (nm, re_, rs)


# ---- the stride-2 down-sampling conv and its data gradient on split operands (csrc/conv_split.hip) ----
def _conv_image(lib, which, d, src, w_off):
    n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
    assert n > 0
    idx = np.empty(n, np.int32)
    assert lib.nsc_conv1d_simage_index(which, C.byref(d), int(w_off), idx.ctypes.data_as(C.c_void_p)) == 0, lib.nsc_last_error()
    img = torch.empty(n, device="cuda")
    assert lib.nsc_gather(src.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, _st()) == 0
    return img


def _down_desc(B, Tin, act=0):
    from nsc_amd._lib import ConvDesc
    return ConvDesc(B=B, Cin=100, Cout=100, Tin=Tin, Tout=Tin // 2, K=9, dil=1, stride=2, padL=3, act=act, res_mode=0, mul_mode=0,
                    out_mode=0, in_up=0, accumulate=0)


@pytest.mark.parametrize("B,Tin,act", [(2, 512, 2), (3, 128, 0), (1, 256, 2), (130, 512, 2)])
def test_split_stride2_conv_forward_and_data_gradient_match_float64_and_the_exact_kernels(lib, B, Tin, act):
    """nsc_conv1d_fwd_simg / nsc_conv1d_dgrad_simg (bf16 matrix cores, split operands) against float64 (same bounds as the exact
    kernels) and against nsc_conv1d_fwd (forward; polyphase data gradient on the gathered W' kernel)."""
    rng = np.random.default_rng(B + Tin)
    Tout = Tin // 2
    w = (0.05 * rng.standard_normal((9, 100, 100))).astype(np.float32)
    bias = (0.1 * rng.standard_normal(100)).astype(np.float32)
    x = rng.standard_normal((B, 100, Tin)).astype(np.float32)
    dy = rng.standard_normal((B, 100, Tout)).astype(np.float32)
    pad = 7                                                   # offset of the kernel in the gathered buffer (not 0: exercises w_off)
    src = dev(np.concatenate([np.zeros(pad, np.float32), w.reshape(-1)]))
    d = _down_desc(B, Tin, act)
    xd, dyd, bd = dev(x), dev(dy), dev(bias)
    # float64 reference
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (3, 4)))
    yr = sum(np.einsum("bit,io->bot", xp[:, :, k:k + 2 * Tout:2], w[k].astype(np.float64)) for k in range(9)) + bias[None, :, None]
    if act == 2:
        yr = np.where(yr > 0, yr, 0.2 * yr)
    dxr = np.zeros((B, 100, Tin + 7))
    for k in range(9):
        dxr[:, :, k:k + 2 * Tout:2] += np.einsum("bot,io->bit", dy.astype(np.float64), w[k].astype(np.float64))
    dxr = dxr[:, :, 3:3 + Tin]
    # split kernels
    y = torch.full((B, 100, Tout), float("nan"), device="cuda")
    img0 = _conv_image(lib, 0, d, src, pad)
    assert lib.nsc_conv1d_fwd_simg(C.byref(d), P(xd), P(img0), P(bd), P(y), _st()) == 0, lib.nsc_last_error()
    dx = torch.full((B, 100, Tin), float("nan"), device="cuda")
    img1 = _conv_image(lib, 1, d, src, pad)
    assert lib.nsc_conv1d_dgrad_simg(C.byref(d), P(dyd), P(img1), P(dx), _st()) == 0, lib.nsc_last_error()
    # exact forward
    ye = torch.empty_like(y)
    assert lib.nsc_conv1d_fwd(C.byref(d), P(xd), src.data_ptr() + 4 * pad, P(bd), None, None, P(ye), _st()) == 0
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy(), yr, what="split stride-2 conv forward")
    assert_close(dx.cpu().numpy(), dxr, what="split stride-2 conv data gradient")
    es = np.sqrt(np.mean((y.cpu().numpy() - yr) ** 2)), np.sqrt(np.mean((ye.cpu().numpy() - yr) ** 2))
    print(f"stride-2 conv forward rms error vs float64: split {es[0]:.3e}, exact {es[1]:.3e}")
    assert es[0] <= 1.5 * es[1] + 1e-7 * np.sqrt(np.mean(yr ** 2))


def test_split_stride2_conv_refuses_other_shapes(lib):
    from nsc_amd._lib import ConvDesc
    d = ConvDesc(B=1, Cin=50, Cout=50, Tin=128, Tout=64, K=9, dil=1, stride=2, padL=3, act=0, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                 accumulate=0)
    assert int(lib.nsc_conv1d_simage_words(0, C.byref(d))) == 0
    d2 = _down_desc(1, 96)                                    # Tout = 48: not a multiple of the 64-column tile
    assert int(lib.nsc_conv1d_simage_words(0, C.byref(d2))) == 0
    t = torch.zeros(16, device="cuda")
    assert lib.nsc_conv1d_fwd_simg(C.byref(d2), P(t), P(t), None, P(t), _st()) == -2


@pytest.mark.parametrize("B,Tin,njobs", [(2, 128, 1), (5, 256, 2), (130, 512, 2), (3, 128, 3)])
def test_split_stride2_conv_weight_gradient_matches_float64_and_the_exact_kernel(lib, B, Tin, njobs):
    """nsc_conv1d_wgrad_split (bf16 matrix cores, split operands; several jobs per launch, accumulating) against float64 and against
    nsc_conv1d_wgrad_ws."""
    from nsc_amd._lib import ConvWgradJob
    rng = np.random.default_rng(B * 7 + Tin + njobs)
    Tout = Tin // 2
    d = _down_desc(B, Tin)
    ws = torch.empty(int(lib.nsc_conv1d_wgrad_split_workspace()), device="cuda")
    keep, jobs, refs = [], [], []
    for q in range(njobs):
        x = rng.standard_normal((B, 100, Tin)).astype(np.float32)
        dy = rng.standard_normal((B, 100, Tout)).astype(np.float32)
        dw0 = (0.5 * rng.standard_normal((9, 100, 100))).astype(np.float32)       # the launch ACCUMULATES into dw / db
        db0 = (0.5 * rng.standard_normal(100)).astype(np.float32)
        xd, dyd, dwd, dbd = dev(x), dev(dy), dev(dw0), dev(db0)
        keep += [xd, dyd, dwd, dbd]
        jobs.append(ConvWgradJob(d, P(xd), P(dyd), P(dwd), P(dbd) if q != 1 else None, 0))      # (job 1: no bias gradient wanted)
        xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (3, 4)))
        dwr = np.stack([np.einsum("bit,bot->io", xp[:, :, k:k + 2 * Tout:2], dy.astype(np.float64)) for k in range(9)])
        refs.append((dwd, dbd, dw0, db0, dwr, dy.astype(np.float64).sum((0, 2)), xd, dyd))
    assert lib.nsc_conv1d_wgrad_split((ConvWgradJob * njobs)(*jobs), njobs, P(ws), ws.numel(), _st()) == 0, lib.nsc_last_error()
    torch.cuda.synchronize()
    for q, (dwd, dbd, dw0, db0, dwr, dbr, xd, dyd) in enumerate(refs):
        assert_close(dwd.cpu().numpy() - dw0, dwr, what=f"split stride-2 conv dW (job {q})")
        if q != 1:
            assert_close(dbd.cpu().numpy() - db0, dbr, what=f"split stride-2 conv db (job {q})")
        else:
            assert np.array_equal(dbd.cpu().numpy(), db0)
    # against the exact kernel (job 0)
    dwd, dbd, dw0, db0, dwr, dbr, xd, dyd = refs[0]
    dwe, dbe = torch.zeros(9, 100, 100, device="cuda"), torch.zeros(100, device="cuda")
    assert lib.nsc_conv1d_wgrad_ws(C.byref(d), P(xd), P(dyd), P(dwe), P(dbe), 0, None, 0, _st()) == 0
    torch.cuda.synchronize()
    es = np.sqrt(np.mean((dwd.cpu().numpy() - dw0 - dwr) ** 2)), np.sqrt(np.mean((dwe.cpu().numpy() - dwr) ** 2))
    print(f"stride-2 conv dW rms error vs float64: split {es[0]:.3e}, exact {es[1]:.3e}")
    assert es[0] <= 1.5 * es[1] + 3e-7 * np.sqrt(np.mean(dwr ** 2))
