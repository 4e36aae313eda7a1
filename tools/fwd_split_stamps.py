"""PROBES build only: per-phase s_memtime stamps of workgroup 0 / waves 0 and 4 of the split-operand block forward (last tile of the
workgroup's chain), next to the exact kernel's."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsc_amd", "libnsc_hip_probes.so")
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
names = ["start", "prologue done", "tile start", "staged", "bar0", "ph1 done", "bar1", "ph2 done", "bar2", "ph3 done", "bar3", "end"]
rng = np.random.default_rng(0)
P = lambda t: t.data_ptr() if t is not None else None
for (B, C_, T, dil, save) in [(128, 100, 512, 1, 1), (128, 100, 512, 2, 1), (128, 50, 512, 2, 1), (1024, 100, 512, 2, 1), (1024, 100, 512, 2, 0)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, C_, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    pd = torch.tensor(np.concatenate([a.reshape(-1) for a in w]), device="cuda")
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    x = torch.randn(B, C_, T, device="cuda")
    out = torch.empty_like(x)
    sv = [torch.empty(B, 20, T, device="cuda") for _ in range(4)] if save else [None] * 4
    for split in (False, True):
        fn_n = lib.nsc_gated_block_simage_words if split else lib.nsc_gated_block_image_floats
        fn_i = lib.nsc_gated_block_simage_index if split else lib.nsc_gated_block_image_index
        n = int(fn_n(0, C_, C_, dil))
        idx = np.empty(n, np.int32)
        _lib.check(fn_i(0, C_, C_, dil, (C.c_long * 8)(*[int(o) for o in offs]), idx.ctypes.data_as(C.c_void_p)), "index")
        img = torch.empty(n, device="cuda")
        _lib.check(lib.nsc_gather(pd.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st), "gather")
        fn = lib.nsc_gated_block_fwd_simg if split else lib.nsc_gated_block_fwd_img
        for _ in range(5):
            _lib.check(fn(P(img), P(x), P(out), *[P(t) for t in sv], B, C_, C_, T, dil, 0, st), "fwd")
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 128)()
        rd = lib.nsc_probe_read_split if split else lib.nsc_probe_read
        rd.argtypes = [C.c_void_p]
        assert rd(buf) == 0
        v, w4 = list(buf)[32:44], list(buf)[96:108]
        print(f"{'split' if split else 'exact'} B={B} C={C_} T={T} dil={dil} save={save}: kernel {v[11]-v[0]} cycles, last tile {v[10]-v[2]}   (wave 0 | wave 4)")
        for i in range(1, 12):
            print(f"  {names[i]:>14}: +{v[i]-v[i-1]:6d} | +{w4[i]-w4[i-1]:6d}")
        if split:
            a, b = list(buf)[:64], list(buf)[64:128]
            print(f"  inside ph2: MFMA loop +{a[44]-a[38]} | +{b[44]-b[38]}, epilogue +{a[39]-a[44]} | +{b[39]-b[44]};  inside ph3: MFMA loop "
                  f"+{a[45]-a[40]} | +{b[45]-b[40]}, vmcnt wait +{a[46]-a[45]} | +{b[46]-b[45]}, epilogue +{a[41]-a[46]} | +{b[41]-b[46]}")
