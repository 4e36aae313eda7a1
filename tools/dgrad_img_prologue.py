"""Prologue of gated_block_dgrad2_kernel on a parameter IMAGE (the engine's path): 'prologue done' stamp of workgroup 0 (waves 0 | 4)
and the launch time, B = 128.  Probes / experiment builds: NSC_LIB=<path> (make -C nsc_amd/csrc exp EXP=1: no first-tile loads in
the prologue; EXP=2: no image loads; EXP=3: neither - timing only, wrong values)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsc_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("NSC_LIB", os.path.join(ROOT, "nsc_amd", "libnsc_hip_probes.so")))
lib = _lib.load()
lib.nsc_probe_read.argtypes = [C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(0)
B = 128
for (C_, T, dil) in [(100, 512, 1), (100, 256, 2), (50, 512, 2)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w1, wl, wr, w9 = f(1, C_, 20), f(15, 20, 20), f(15, 20, 20), f(9, 20, C_)
    wt = [np.ascontiguousarray(w[::-1].transpose(0, 2, 1)) for w in (w1, wl, wr, w9)]
    td = dev(np.concatenate([w.reshape(-1) for w in wt]))
    toffs = np.concatenate([[0], np.cumsum([w.size for w in wt])[:-1]]).astype(np.int64)
    n = int(lib.nsc_gated_block_image_floats(1, C_, C_, dil))
    idx = np.empty(n, np.int32)
    assert lib.nsc_gated_block_image_index(1, C_, C_, dil, (C.c_long * 4)(*[int(o) for o in toffs]), idx.ctypes.data_as(C.c_void_p)) == 0
    img = torch.empty(n, device="cuda")
    assert lib.nsc_gather(td.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st) == 0
    x, dy = torch.randn(B, C_, T, device="cuda"), torch.randn(B, C_, T, device="cuda")
    h, lin = torch.randn(B, 20, T, device="cuda"), torch.randn(B, 20, T, device="cuda")
    th = torch.tanh(torch.randn(B, 20, T, device="cuda"))
    dx, da, dz1 = torch.empty_like(x), torch.empty(B, 40, T, device="cuda"), torch.empty(B, 20, T, device="cuda")
    def run():
        _lib.check(lib.nsc_gated_block_dgrad_img(img.data_ptr(), x.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), dy.data_ptr(),
                                                 dx.data_ptr(), da.data_ptr(), da.data_ptr() + 4 * 20 * T, dz1.data_ptr(), B, C_, C_, T, dil,
                                                 2, 40, st), "dgrad_img")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert lib.nsc_probe_read(buf) == 0
    v, w = list(buf)[:18], list(buf)[64:82]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C_} T={T} dil={dil}: prologue done +{v[1]-v[0]} | +{w[1]-w[0]} cycles; kernel total {v[17]-v[0]} cycles; launch {1e3 * e0.elapsed_time(e1) / 20:.1f} us")
