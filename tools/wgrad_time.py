"""Times nsc_conv1d_wgrad_ws (slab) vs nsc_conv1d_wgrad (atomics) on the codec's per-conv weight-gradient shapes."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])      # A/B against another build
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = 128
# (Cin, Cout, Tin, K, dil, stride, padL)
cases = [(1, 100, 512, 55, 1, 1, 27), (1, 100, 256, 55, 1, 1, 27), (100, 100, 256, 1, 1, 1, 0), (20, 100, 256, 9, 1, 1, 4),
         (20, 20, 256, 15, 2, 1, 14), (1, 20, 256, 1, 1, 1, 0), (1, 50, 512, 55, 1, 1, 27), (100, 100, 512, 9, 1, 2, 3)]
for (Cin, Cout, T, K, dil, s, padL) in cases:
    Tout = -(-T // s)
    d = ConvDesc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=Tout, K=K, dil=dil, stride=s, padL=padL, act=0, res_mode=0, mul_mode=0,
                 out_mode=0, in_up=0, accumulate=0)
    x = torch.randn(B, Cin, T, device="cuda"); dz = torch.randn(B, Cout, Tout, device="cuda")
    dw = torch.zeros(K, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
    nws = lib.nsc_conv1d_wgrad_workspace(C.byref(d))
    ws = torch.empty(nws, device="cuda")
    def slab():
        _lib.check(lib.nsc_conv1d_wgrad_ws(C.byref(d), x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), nws, st), "w")
    def atom():
        _lib.check(lib.nsc_conv1d_wgrad(C.byref(d), x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, st), "w")
    res = []
    for fn in (slab, atom):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / 20)
    fl = 2.0 * B * Tout * K * Cin * Cout
    print(f"Cin={Cin} Cout={Cout} T={T} K={K} s={s}: slab {res[0]:7.1f} us ({fl / res[0] / 1e6:5.1f} TF)  atomics {res[1]:7.1f} us  ws {nws * 4 / 1e6:.1f} MB")
