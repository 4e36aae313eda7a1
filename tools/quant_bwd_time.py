"""nsc_quantize_bwd / nsc_quantize_fwd at the TRAINING shapes (B = 128: the codec quantizer L = 256 x 32 bins, the LSF quantizer L = 16 x 256
bins), replayed from a hipGraph of 50 launches: 7-10 us each, and the same with the global atomics of the epilogue compiled out (round 6:
9.86 -> 9.84 us and 9.56 -> 8.16 us) - launch-to-launch latency, not contention.   [NSC_LIB=<build>] python tools/quant_bwd_time.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsc_amd import _lib
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
dev = "cuda"
B = 128
for L, nb in ((256, 32), (16, 256)):
    code = torch.tanh(torch.randn(B, L, 1, device=dev)); alpha = torch.tensor([-20.0], device=dev); bins = torch.linspace(-1, 1, nb, device=dev)
    dout = torch.randn(B, L, 1, device=dev); gh = torch.randn(nb, device=dev)
    dcode = torch.empty_like(code); dalpha = torch.zeros(1, device=dev); dbins = torch.zeros(nb, device=dev)
    outq = torch.empty_like(code); qv = torch.empty(B, device=dev); hist = torch.zeros(nb, device=dev)
    s = torch.cuda.Stream()
    def bwd():
        _lib.check(lib.nsc_quantize_bwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, 1, B, L, nb, dout.data_ptr(), None, 10.0,
                                        gh.data_ptr(), 0.3, 0, dcode.data_ptr(), dalpha.data_ptr(), dbins.data_ptr(), s.cuda_stream), "bwd")
    def fwd():
        _lib.check(lib.nsc_quantize_fwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, 1, B, L, nb, None, outq.data_ptr(), qv.data_ptr(),
                                        hist.data_ptr(), s.cuda_stream), "fwd")
    for name, fn in (("quantize_bwd", bwd), ("quantize_fwd", fwd)):
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} B={B} L={L} nb={nb}: {1e3 * e0.elapsed_time(e1) / 500:.2f} us per launch")
