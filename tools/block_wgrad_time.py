"""Times nsc_gated_block_wgrad_batch on the headline step's two launches: the C = 100 blocks (T = 512 dil 1,2,1,2 and T = 256
dil 2,1,2,1 per codec, two codecs) and the C = 50 blocks.  NSC_LIB=<path> times another build."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import BlockWgradJob
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
st = torch.cuda.current_stream().cuda_stream
B = 128
for name, shapes in (("C=100 x10", [(100, 512, 1), (100, 512, 2)] * 2 + [(100, 256, 2), (100, 256, 1)] * 3),
                     ("C=50 x8", [(50, 512, 1), (50, 512, 2)] * 4)):
    keep, jobs, fl = [], [], 0.0
    for (Cc, T, dil) in shapes:
        x, dy = torch.randn(B, Cc, T, device="cuda"), torch.randn(B, Cc, T, device="cuda")
        h, g, dz1 = (torch.randn(B, 20, T, device="cuda") for _ in range(3))
        da = torch.randn(B, 40, T, device="cuda")
        n = Cc * 20 + 20 + 2 * (15 * 20 * 20 + 20) + 9 * 20 * Cc + Cc
        gr = torch.zeros(n, device="cuda")
        keep += [x, dy, h, g, dz1, da, gr]
        jobs.append(BlockWgradJob(x.data_ptr(), h.data_ptr(), g.data_ptr(), dy.data_ptr(), da.data_ptr(), dz1.data_ptr(), gr.data_ptr(),
                                  Cc, T, dil, Cc))
        fl += 2.0 * B * T * (Cc * 20 + 2 * 15 * 20 * 20 + 9 * 20 * Cc)
    arr = (BlockWgradJob * len(jobs))(*jobs)
    nws = lib.nsc_gated_block_wgrad_batch_workspace(100)
    ws = torch.empty(nws, device="cuda")
    run = lambda: _lib.check(lib.nsc_gated_block_wgrad_batch(arr, len(jobs), B, 20, 9, ws.data_ptr(), nws, st), "wgrad_batch")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    print(f"{name}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (incl. the slab reduce)")
