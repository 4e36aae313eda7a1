"""Times nsc_gated_block_fwd on the codec's block shapes (NSC_BLOCK_FWD_V1=1 selects the per-tile kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])      # A/B against another build
lib = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for (B, C, T, dil, save) in [(1, 100, 256, 2, 0), (128, 100, 256, 2, 0), (1024, 100, 256, 2, 0), (128, 100, 512, 1, 1), (128, 100, 256, 2, 1), (128, 50, 512, 2, 1), (4096, 100, 256, 2, 0), (4096, 50, 512, 1, 0)]:
    x = torch.randn(B, C, T, device=dev)
    w1 = torch.randn(1, C, 20, device=dev) * 0.1; b1 = torch.zeros(20, device=dev)
    wl = torch.randn(15, 20, 20, device=dev) * 0.05; wr = torch.randn(15, 20, 20, device=dev) * 0.05
    bl = torch.zeros(20, device=dev); br = torch.zeros(20, device=dev)
    w9 = torch.randn(9, 20, C, device=dev) * 0.05; b9 = torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    sv = [torch.empty(B, 20, T, device=dev) for _ in range(4)] if save else [None] * 4
    p = lambda t: t.data_ptr() if t is not None else None
    def run():
        _lib.check(lib.nsc_gated_block_fwd(p(x), p(w1), p(b1), p(wl), p(bl), p(wr), p(br), p(w9), p(b9), p(out), p(sv[0]), p(sv[1]),
                                           p(sv[2]), p(sv[3]), B, C, T, 20, 9, dil, 0, st), "blk")
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
    print(f"B={B} C={C} T={T} dil={dil} save={save}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
