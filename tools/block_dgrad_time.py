"""Times nsc_gated_block_dgrad on the codec's block shapes (NSC_BLOCK_DGRAD_V1=1 selects the per-tile kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])      # A/B against another build
lib = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
shapes = [(1, 100, 256, 2), (128, 100, 512, 1), (128, 100, 256, 2), (128, 50, 512, 2), (1024, 100, 256, 2), (1024, 50, 512, 1)]
for (B, C, T, dil) in shapes:
    x = torch.randn(B, C, T, device=dev); dy = torch.randn(B, C, T, device=dev)
    h, lin, th = (torch.randn(B, 20, T, device=dev) for _ in range(3)); th = torch.tanh(th)
    wt1 = torch.randn(1, 20, C, device=dev) * 0.1
    wtl = torch.randn(15, 20, 20, device=dev) * 0.05; wtr = torch.randn(15, 20, 20, device=dev) * 0.05
    wt9 = torch.randn(9, C, 20, device=dev) * 0.05
    dx = torch.empty_like(x); da = torch.empty(B, 40, T, device=dev); dz1 = torch.empty(B, 20, T, device=dev)
    p = lambda t: t.data_ptr()
    def run():
        _lib.check(lib.nsc_gated_block_dgrad(p(x), p(h), p(lin), p(th), p(dy), p(wt1), p(wtl), p(wtr), p(wt9), p(dx), p(da), p(dz1),
                                             B, C, T, 20, 9, dil, 2, st), "dgrad")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    fl = 2.0 * B * T * (C * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C)
    print(f"B={B} C={C} T={T} dil={dil}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
