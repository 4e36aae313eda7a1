"""Times nsc_conv1d_fwd on the per-conv shapes of the codec (forward and data-gradient forms)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])      # A/B against another build
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("B", 128))
# (name, Cin, Cout, Tin, Tout, K, dil, stride, padL, in_up)
cases = [("down fwd", 100, 100, 512, 256, 9, 1, 2, 3, 0), ("down dgrad", 100, 100, 256, 512, 9, 1, 1, 4, 1),
         ("pointwise", 100, 100, 256, 256, 1, 1, 1, 0, 0), ("in conv", 1, 100, 512, 512, 55, 1, 1, 27, 0),
         ("dec in conv", 1, 100, 256, 256, 55, 1, 1, 27, 0), ("k9 20->100", 20, 100, 256, 256, 9, 1, 1, 4, 0),
         ("k15 20->20", 20, 20, 256, 256, 15, 2, 1, 14, 0), ("k9T 100->20", 100, 20, 256, 256, 9, 1, 1, 4, 0),
         ("out dgrad 1->50", 1, 50, 512, 512, 55, 1, 1, 27, 0)]
for (name, Cin, Cout, Tin, Tout, K, dil, s, padL, in_up) in cases:
    d = ConvDesc(B=B, Cin=Cin, Cout=Cout, Tin=Tin, Tout=Tout, K=K, dil=dil, stride=s, padL=padL, act=2, res_mode=0, mul_mode=0,
                 out_mode=0, in_up=in_up, accumulate=0)
    x = torch.randn(B, Cin, Tin, device="cuda"); w = torch.randn(K, Cin, Cout, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda"); y = torch.empty(B, Cout, Tout, device="cuda")
    def run():
        _lib.check(lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), st), "c")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    fl = 2.0 * B * Tout * K * Cin * Cout / (2 if in_up else 1)
    print(f"{name:16s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s   {4.0 * B * (Cin * Tin + Cout * Tout) / us / 1e3:7.1f} GB/s")
