"""A/B timing of the gated-block forward on parameter images: exact fp32 MFMA (nsc_gated_block_fwd_img) against the bf16 matrix cores on
split operands (nsc_gated_block_fwd_simg), single launches and pair launches, on the headline step's block shapes.
   python tools/block_split_time.py            (NSC_LIB=path: another build)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib

if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)


def image(split, C_, Cin, dil, pd, offs):
    fn_n = lib.nsc_gated_block_simage_words if split else lib.nsc_gated_block_image_floats
    fn_i = lib.nsc_gated_block_simage_index if split else lib.nsc_gated_block_image_index
    n = int(fn_n(0, C_, Cin, dil))
    idx = np.empty(n, np.int32)
    _lib.check(fn_i(0, C_, Cin, dil, (C.c_long * len(offs))(*[int(o) for o in offs]), idx.ctypes.data_as(C.c_void_p)), "index")
    img = torch.empty(n, device="cuda")
    _lib.check(lib.nsc_gather(pd.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st), "gather")
    return img


def params(Cin, C_):
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, Cin, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    flat = np.concatenate([a.reshape(-1) for a in w])
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    return torch.tensor(flat, device="cuda"), offs


def timeit(run, n=30):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


P = lambda t: t.data_ptr() if t is not None else None
print("single launches (us | TFLOP/s algorithmic):")
for (B, C_, Cin, T, dil, save) in [(128, 100, 100, 512, 1, 1), (128, 100, 100, 512, 2, 1), (128, 100, 100, 256, 1, 1), (128, 100, 100, 256, 2, 1),
                                   (128, 50, 50, 512, 1, 1), (128, 50, 50, 512, 2, 1), (128, 100, 1, 256, 1, 1), (1024, 100, 100, 256, 2, 0),
                                   (4096, 100, 100, 256, 2, 0), (4096, 50, 50, 512, 1, 0)]:
    pd, offs = params(Cin, C_)
    x = torch.randn(B, Cin, T, device="cuda")
    out = torch.empty(B, C_, T, device="cuda")
    sv = [torch.empty(B, 20, T, device="cuda") for _ in range(4)] if save else [None] * 4
    fl = 2.0 * B * T * (Cin * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C_)
    row = []
    for split in (False, True):
        img = image(split, C_, Cin, dil, pd, offs)
        fn = lib.nsc_gated_block_fwd_simg if split else lib.nsc_gated_block_fwd_img
        us = timeit(lambda: _lib.check(fn(P(img), P(x), P(out), *[P(t) for t in sv], B, C_, Cin, T, dil, 0, st), "fwd"))
        row.append(us)
    print(f"  B={B:5d} C={C_:3d} Cin={Cin:3d} T={T} dil={dil} save={save}: exact {row[0]:7.1f} us {fl / row[0] / 1e6:6.1f} | split {row[1]:7.1f} us "
          f"{fl / row[1] / 1e6:6.1f} | x{row[0] / row[1]:.2f}")

print("pair launches (dil 1 + dil 2):")
nfl = int(lib.nsc_gated_block_pair_flag_ints())
tmo = torch.zeros(4, dtype=torch.int32, device="cuda")
for (B, C_, Cin0, T, save) in [(128, 100, 100, 512, 1), (128, 100, 100, 256, 1), (128, 50, 50, 512, 1), (128, 100, 1, 256, 1), (4096, 100, 100, 256, 0)]:
    ps = [params(Cin0, C_), params(C_, C_)]
    x = torch.randn(B, Cin0, T, device="cuda")
    o0, o1 = torch.empty(B, C_, T, device="cuda"), torch.empty(B, C_, T, device="cuda")
    s0 = [torch.empty(B, 20, T, device="cuda") for _ in range(4)] if save else [None] * 4
    s1 = [torch.empty(B, 20, T, device="cuda") for _ in range(4)] if save else [None] * 4
    flags = torch.zeros(nfl, dtype=torch.int32, device="cuda")
    fl = 2.0 * B * T * ((Cin0 + C_) * 20 + 2 * (2 * 15 * 20 * 20 + 9 * 20 * C_))
    row = []
    for split in (False, True):
        i0 = image(split, C_, Cin0, 1, *ps[0])
        i1 = image(split, C_, C_, 2, *ps[1])
        fn = lib.nsc_gated_block_pair_fwd_simg if split else lib.nsc_gated_block_pair_fwd_img

        def run():
            flags.zero_()
            _lib.check(fn(P(i0), P(i1), P(x), P(o0), *[P(t) for t in s0], P(o1), *[P(t) for t in s1], B, C_, Cin0, T, 1, P(flags), P(tmo), st), "pair")
        row.append(timeit(run))
    print(f"  B={B:5d} C={C_:3d} Cin0={Cin0:3d} T={T} save={save}: exact {row[0]:7.1f} us {fl / row[0] / 1e6:6.1f} | split {row[1]:7.1f} us "
          f"{fl / row[1] / 1e6:6.1f} | x{row[0] / row[1]:.2f}   time-outs {int(tmo[0])}")
