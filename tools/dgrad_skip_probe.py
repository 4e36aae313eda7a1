"""Probes library: wall time of nsc_gated_block_dgrad with one phase skipped (NSC_DGRAD2_SKIP bit mask: 1 = k9 gradient,
2 = k15 gradient, 4 = 1x1 gradient, 8 = next-tile prefetch) at a long-chain shape - what each phase costs in the pipeline."""
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
for (B, C, T, dil) in [(1024, 100, 512, 2), (128, 100, 512, 2)]:
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
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    ntile = B * T // 64 // 256
    print(f"  B={B}: {us:8.1f} us  = {us / ntile:6.2f} us per tile per workgroup ({ntile} tiles)")
''' % (ROOT, ROOT)
for skip, name in ((0, "nothing skipped"), (1, "k9 gradient skipped"), (2, "k15 gradient skipped"), (3, "both skipped"), (4, "1x1 gradient skipped"), (7, "all three MFMA phases skipped")):
    print(f"NSC_DGRAD2_SKIP={skip} ({name})", flush=True)
    env = dict(os.environ, NSC_DGRAD2_SKIP=str(skip))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(r.stdout, end="", flush=True)
    if r.returncode:
        print(r.stderr[-800:])
