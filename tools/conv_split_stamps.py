"""PROBES build only: phase stamps (s_memtime) of workgroup 0 / waves 0 and 4 of the split-operand stride-2 conv kernels, last tile of the workgroup."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc
_lib.LIB_PATH = os.environ.get("NSC_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsc_amd", "libnsc_hip_probes.so")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
names = ["tile start", "staged", "barrier", "GEMM done", "epilogue", "barrier"]
for B in (128, 1024):
    d = ConvDesc(B=B, Cin=100, Cout=100, Tin=512, Tout=256, K=9, dil=1, stride=2, padL=3, act=2, res_mode=0, mul_mode=0, out_mode=0, in_up=0, accumulate=0)
    w = torch.randn(9 * 100 * 100, device="cuda") * 0.05
    bias = torch.zeros(100, device="cuda")
    x, y = torch.randn(B, 100, 512, device="cuda"), torch.empty(B, 100, 256, device="cuda")
    dy, dx = torch.randn(B, 100, 256, device="cuda"), torch.empty(B, 100, 512, device="cuda")
    for which in (0, 1):
        n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
        idx = np.empty(n, np.int32)
        _lib.check(lib.nsc_conv1d_simage_index(which, C.byref(d), 0, idx.ctypes.data_as(C.c_void_p)), "index")
        img = torch.empty(n, device="cuda")
        _lib.check(lib.nsc_gather(P(w), P(torch.tensor(idx, device="cuda")), P(img), n, st), "gather")
        for _ in range(4):
            if which == 0:
                _lib.check(lib.nsc_conv1d_fwd_simg(C.byref(d), P(x), P(img), P(bias), P(y), st), "fwd")
            else:
                _lib.check(lib.nsc_conv1d_dgrad_simg(C.byref(d), P(dy), P(img), P(dx), st), "dgrad")
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 128)()
        lib.nsc_probe_read_conv_split.argtypes = [C.c_void_p]
        assert lib.nsc_probe_read_conv_split(buf) == 0
        v, w4 = list(buf)[0:6], list(buf)[64:70]
        print(f"{'forward' if which == 0 else 'data gradient'} B={B}: last tile {v[5] - v[0]} cycles   (wave 0 | wave 4)")
        for i in range(1, 6):
            print(f"  {names[i]:>12}: +{v[i] - v[i - 1]:6d} | +{w4[i] - w4[i - 1]:6d}")
