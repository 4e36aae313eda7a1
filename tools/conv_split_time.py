"""A/B per launch: the stride-2 down-sampling conv (forward, polyphase data gradient) on the exact fp32 matrix instruction
(nsc_conv1d_fwd) against the bf16 matrix cores on split operands (nsc_conv1d_fwd_simg / _dgrad_simg).   python tools/conv_split_time.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
P = lambda t: t.data_ptr()


def timeit(run, n=50):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B, Tin in [(128, 512), (256, 512), (1024, 512), (128, 256)]:
    Tout = Tin // 2
    d = ConvDesc(B=B, Cin=100, Cout=100, Tin=Tin, Tout=Tout, K=9, dil=1, stride=2, padL=3, act=2, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                 accumulate=0)
    w = torch.tensor((0.05 * rng.standard_normal(9 * 100 * 100)).astype(np.float32), device="cuda")
    bias = torch.zeros(100, device="cuda")
    x, y = torch.randn(B, 100, Tin, device="cuda"), torch.empty(B, 100, Tout, device="cuda")
    dy, dx = torch.randn(B, 100, Tout, device="cuda"), torch.empty(B, 100, Tin, device="cuda")
    imgs = []
    for which in (0, 1):
        n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
        idx = np.empty(n, np.int32)
        _lib.check(lib.nsc_conv1d_simage_index(which, C.byref(d), 0, idx.ctypes.data_as(C.c_void_p)), "index")
        img = torch.empty(n, device="cuda")
        _lib.check(lib.nsc_gather(P(w), P(torch.tensor(idx, device="cuda")), P(img), n, st), "gather")
        imgs.append(img)
    # the exact polyphase data gradient reads W'[t'][o][2 ci + p]
    wnp = w.cpu().numpy().reshape(9, 100, 100)
    wp = np.zeros((5, 100, 200), np.float32)
    for tp in range(5):
        for par in range(2):
            k = 7 - 2 * tp + par
            if 0 <= k < 9:
                wp[tp, :, par::2] = wnp[k].T
    wpd = torch.tensor(wp, device="cuda")
    dd = ConvDesc(B=B, Cin=100, Cout=200, Tin=Tout, Tout=Tout, K=5, dil=1, stride=1, padL=2, act=0, res_mode=0, mul_mode=0, out_mode=1, in_up=0,
                  accumulate=0)
    fl = 2.0 * B * Tout * 9 * 100 * 100
    t_fe = timeit(lambda: _lib.check(lib.nsc_conv1d_fwd(C.byref(d), P(x), P(w), P(bias), None, None, P(y), st), "fwd"))
    t_fs = timeit(lambda: _lib.check(lib.nsc_conv1d_fwd_simg(C.byref(d), P(x), P(imgs[0]), P(bias), P(y), st), "fwd simg"))
    t_de = timeit(lambda: _lib.check(lib.nsc_conv1d_fwd(C.byref(dd), P(dy), P(wpd), None, None, None, P(dx), st), "dgrad"))
    dxe = dx.clone()
    t_ds = timeit(lambda: _lib.check(lib.nsc_conv1d_dgrad_simg(C.byref(d), P(dy), P(imgs[1]), P(dx), st), "dgrad simg"))
    err = float((dx - dxe).abs().max() / dxe.abs().max())
    # weight gradient: two jobs per launch, as in the 2-codec step
    from nsc_amd._lib import ConvWgradJob
    dw = [torch.zeros(9, 100, 100, device="cuda") for _ in range(2)]
    db = [torch.zeros(100, device="cuda") for _ in range(2)]
    jobs = (ConvWgradJob * 2)(*[ConvWgradJob(d, P(x), P(dy), P(dw[q]), P(db[q]), 0) for q in range(2)])
    wsb = torch.empty(int(lib.nsc_conv1d_wgrad_batch_workspace(jobs, 2)), device="cuda")
    wss = torch.empty(int(lib.nsc_conv1d_wgrad_split_workspace()), device="cuda")
    t_we = timeit(lambda: _lib.check(lib.nsc_conv1d_wgrad_batch(jobs, 2, P(wsb), wsb.numel(), st), "wgrad batch"), n=20)
    t_ws = timeit(lambda: _lib.check(lib.nsc_conv1d_wgrad_split(jobs, 2, P(wss), wss.numel(), st), "wgrad split"), n=20)
    print(f"B={B:5d} Tin={Tin}: forward exact {t_fe:6.1f} us {fl / t_fe / 1e6:6.1f} TF | split {t_fs:6.1f} us {fl / t_fs / 1e6:6.1f} TF | x{t_fe / t_fs:.2f}    "
          f"data gradient exact {t_de:6.1f} us | split {t_ds:6.1f} us {fl / t_ds / 1e6:6.1f} TF | x{t_de / t_ds:.2f}  (split vs exact dx: {err:.1e})    "
          f"weight gradient, 2 jobs: exact {t_we:6.1f} us | split {t_ws:6.1f} us {2 * fl / t_ws / 1e6:6.1f} TF | x{t_we / t_ws:.2f}")
