"""Times nsc_conv1d_cout1_fwd on the model's Cout = 1 shapes (NSC_COUT1_V1=1 selects the two-LDS-reads-per-FMA kernel)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (B, Cin, T) in [(128, 100, 256), (128, 50, 512), (128, 100, 512), (4096, 100, 256), (4096, 50, 512)]:
    d = ConvDesc(B=B, Cin=Cin, Cout=1, Tin=T, Tout=T, K=55, dil=1, stride=1, padL=27, act=0, res_mode=0, mul_mode=0, out_mode=0,
                 in_up=0, accumulate=0)
    x = torch.randn(B, Cin, T, device="cuda"); w = torch.randn(55, Cin, 1, device="cuda"); b = torch.zeros(1, device="cuda")
    y = torch.empty(B, 1, T, device="cuda")
    def run():
        _lib.check(lib.nsc_conv1d_cout1_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), st), "c")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f"B={B} Cin={Cin} T={T}: {us:8.1f} us  {2.0 * B * T * 55 * Cin / us / 1e6:6.1f} TFLOP/s  {4.0 * B * Cin * T / us / 1e3:7.1f} GB/s")
