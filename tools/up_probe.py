"""Time the fused up-sampling stage kernels at the bench shape (B = 128, C = 100, T = 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
import os as _os
if _os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = _os.path.abspath(_os.environ["NSC_LIB"])
lib = _lib.load()
B, C_, T = 128, 100, 256
x = torch.randn(B, C_, T, device="cuda"); wd = torch.randn(9, C_, device="cuda"); wp = torch.randn(C_, C_, device="cuda") * 0.1
bias = torch.randn(C_, device="cuda"); dwo = torch.empty_like(x); y = torch.empty(B, C_ // 2, 2 * T, device="cuda")
dz = torch.randn_like(y); dzp, ddw, dx = (torch.empty_like(x) for _ in range(3))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("fwd (dwo saved) %.1f us" % timeit(lambda: lib.nsc_upsample_fwd(p(x), p(wd), p(wp), p(bias), p(dwo), p(y), B, C_, T, 9, 2, st)))
print("fwd (inference) %.1f us" % timeit(lambda: lib.nsc_upsample_fwd(p(x), p(wd), p(wp), p(bias), None, p(y), B, C_, T, 9, 2, st)))
print("bwd             %.1f us" % timeit(lambda: lib.nsc_upsample_bwd(p(dz), p(wd), p(wp), p(dzp), p(ddw), p(dx), B, C_, T, 9, st)))
