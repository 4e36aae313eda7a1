"""A/B timing of the gated-block data gradient on parameter images: exact fp32 MFMA (nsc_gated_block_dgrad_img) against the split-operand
kernel (nsc_gated_block_dgrad_simg); with the probes library (NSC_LIB_PATH=.../libnsc_hip_probes.so) also the split kernel's phase stamps."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
P = lambda t: t.data_ptr()
probes = hasattr(lib, "nsc_probe_read_split")
names = ["start", "prologue", "tile start", "staged", "bar", "k9 grad", "bar", "GLU", "bar", "k15 grad", "bar", "lrelu'", "bar", "1x1 + out", "bar", "end"]
for (B, C_, T, dil) in [(128, 100, 512, 1), (128, 100, 512, 2), (128, 100, 256, 2), (128, 50, 512, 2), (1024, 100, 256, 2)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, C_, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    pd = torch.tensor(np.concatenate([a.reshape(-1) for a in w]), device="cuda")
    wt = [np.ascontiguousarray(w[i][::-1].transpose(0, 2, 1)) for i in (0, 2, 4, 6)]
    td = torch.tensor(np.concatenate([a.reshape(-1) for a in wt]), device="cuda")
    toffs = np.concatenate([[0], np.cumsum([a.size for a in wt])[:-1]]).astype(np.int64)
    x, dy = torch.randn(B, C_, T, device="cuda"), torch.randn(B, C_, T, device="cuda")
    h, lin = torch.randn(B, 20, T, device="cuda"), torch.randn(B, 20, T, device="cuda")
    th = torch.tanh(torch.randn(B, 20, T, device="cuda"))
    dx, da, dz1 = torch.empty_like(x), torch.empty(B, 40, T, device="cuda"), torch.empty(B, 20, T, device="cuda")
    fl = 2.0 * B * T * (C_ * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C_)
    row = []
    for split in (False, True):
        fn_n = lib.nsc_gated_block_simage_words if split else lib.nsc_gated_block_image_floats
        fn_i = lib.nsc_gated_block_simage_index if split else lib.nsc_gated_block_image_index
        src, o = (pd, offs) if split else (td, toffs)
        n = int(fn_n(1, C_, C_, dil))
        idx = np.empty(n, np.int32)
        _lib.check(fn_i(1, C_, C_, dil, (C.c_long * len(o))(*[int(v) for v in o]), idx.ctypes.data_as(C.c_void_p)), "index")
        img = torch.empty(n, device="cuda")
        _lib.check(lib.nsc_gather(src.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st), "gather")
        fn = lib.nsc_gated_block_dgrad_simg if split else lib.nsc_gated_block_dgrad_img
        run = lambda: _lib.check(fn(P(img), P(x), P(h), P(lin), P(th), P(dy), P(dx), P(da), P(da) + 80 * T, P(dz1), B, C_, C_, T, dil, 2, 40, st), "dgrad")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        row.append(1e3 * e0.elapsed_time(e1) / 20)
    print(f"B={B:5d} C={C_:3d} T={T} dil={dil}: exact {row[0]:7.1f} us {fl / row[0] / 1e6:6.1f} TF | split {row[1]:7.1f} us {fl / row[1] / 1e6:6.1f} TF | x{row[0] / row[1]:.2f}")
    if probes:
        buf = (C.c_ulonglong * 128)()
        lib.nsc_probe_read_split.argtypes = [C.c_void_p]
        assert lib.nsc_probe_read_split(buf) == 0
        v, w4 = list(buf)[0:16], list(buf)[64:80]
        print(f"   split kernel {v[15] - v[0]} cycles, last tile {v[14] - v[2]}   (wave 0 | wave 4)")
        for i in range(1, 16):
            print(f"   {names[i]:>12}: +{v[i] - v[i - 1]:6d} | +{w4[i] - w4[i - 1]:6d}")
