"""Probes library: wall time of nsc_gated_block_fwd with one phase skipped (NSC_FWD2_SKIP bit mask: 1 = 1x1, 2 = k15 gates,
4 = k9, 8 = next-tile prefetch) - what each phase costs in the pipeline; with and without the saved activations."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from nsc_amd import _lib
_lib.LIB_PATH = os.path.join(%r, "nsc_amd", "libnsc_hip_probes.so")
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])      # A/B against another build
lib = _lib.load()
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None
for (B, C, T, dil, save) in [(1024, 100, 512, 2, 1), (1024, 100, 512, 2, 0), (128, 100, 512, 2, 1)]:
    x = torch.randn(B, C, T, device=dev)
    w1 = torch.randn(1, C, 20, device=dev) * 0.1; b1 = torch.zeros(20, device=dev)
    wl = torch.randn(15, 20, 20, device=dev) * 0.05; wr = torch.randn(15, 20, 20, device=dev) * 0.05
    bl = torch.zeros(20, device=dev); br = torch.zeros(20, device=dev)
    w9 = torch.randn(9, 20, C, device=dev) * 0.05; b9 = torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    sv = [torch.empty(B, 20, T, device=dev) for _ in range(4)] if save else [None] * 4
    def run():
        _lib.check(lib.nsc_gated_block_fwd(p(x), p(w1), p(b1), p(wl), p(bl), p(wr), p(br), p(w9), p(b9), p(out), p(sv[0]), p(sv[1]),
                                           p(sv[2]), p(sv[3]), B, C, T, 20, 9, dil, 0, st), "blk")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    ntile = B * T // 64 // 256
    print(f"  B={B} save={save}: {us:8.1f} us  = {us / ntile:6.2f} us per tile per workgroup ({ntile} tiles)")
''' % (ROOT, ROOT)
for skip, name in ((0, "nothing skipped"), (1, "1x1 skipped"), (2, "k15 gates skipped"), (4, "k9 skipped"), (7, "all three MFMA phases skipped"), (15, "MFMA phases and prefetch skipped")):
    print(f"NSC_FWD2_SKIP={skip} ({name})", flush=True)
    env = dict(os.environ, NSC_FWD2_SKIP=str(skip))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(r.stdout, end="", flush=True)
    if r.returncode:
        print(r.stderr[-800:])
