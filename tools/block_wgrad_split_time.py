"""A/B timing of the batched block weight gradients of the headline step (12 jobs at C = 100, 4 at C = 50, B = 128): exact fp32 MFMA
against the bf16 matrix cores on split operands."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import BlockWgradJob
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("B", "128"))
shapes = [(100, 100, 512, 1), (100, 100, 512, 2), (100, 100, 256, 1), (100, 100, 256, 2), (100, 1, 256, 1), (100, 100, 256, 2)] * 2 + \
         [(50, 50, 512, 1), (50, 50, 512, 2)] * 2
keep, jobs = [], []
fl = 0.0
for (C_, Cx, T, dil) in shapes:
    t = {k: torch.randn(*s, device="cuda") for k, s in dict(x=(B, Cx, T), h=(B, 20, T), g=(B, 20, T), dy=(B, C_, T), da=(B, 40, T), dz1=(B, 20, T)).items()}
    n = Cx * 20 + 20 + 2 * 6020 + 180 * C_ + C_
    out = torch.zeros(n, device="cuda")
    keep.append((t, out))
    jobs.append(BlockWgradJob(t["x"].data_ptr(), t["h"].data_ptr(), t["g"].data_ptr(), t["dy"].data_ptr(), t["da"].data_ptr(), t["dz1"].data_ptr(),
                              out.data_ptr(), C_, T, dil, Cx if Cx != C_ else 0))
    fl += 2.0 * B * T * (Cx * 20 + 2 * 6000 + 180 * C_)
lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
nws = 2 * lib.nsc_gated_block_wgrad_batch_workspace(100)
ws = torch.empty(nws, device="cuda")
arr = (BlockWgradJob * len(jobs))(*jobs)
for name, fn in (("exact", lib.nsc_gated_block_wgrad_batch), ("split", lib.nsc_gated_block_wgrad_batch_split)):
    for _ in range(3):
        _lib.check(fn(arr, len(jobs), B, 20, 9, ws.data_ptr(), nws, st), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        _lib.check(fn(arr, len(jobs), B, 20, 9, ws.data_ptr(), nws, st), name)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    print(f"{name}: {us:8.1f} us per step's block weight gradients (2 launches + 1 reduce), {fl / us / 1e6:6.1f} TFLOP/s algorithmic")
