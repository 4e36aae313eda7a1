"""Phase stamps of bb_gemm_kernel (csrc/block_bwd_split.hip) with the probes library: NSC_LIB_PATH=nsc_amd/libnsc_hip_probes.so.
Stamps of workgroup 0, waves 0 and 4, LAST tile of the launch: tile start | staged | barrier | GEMM | combine | epilogue | barrier."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
P = lambda t: t.data_ptr()
names = ["tile start", "staged", "barrier", "GEMM", "combine", "epilogue", "barrier"]
lib.nsc_probe_read_bb.argtypes = [C.c_void_p]
for (B, C_, T, dil) in [(128, 100, 512, 1), (128, 100, 256, 2), (128, 50, 512, 2)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, C_, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    pd = torch.tensor(np.concatenate([a.reshape(-1) for a in w]), device="cuda")
    n = int(lib.nsc_gated_block_simage_words(2, C_, C_, dil))
    idx = np.empty(n, np.int32)
    _lib.check(lib.nsc_gated_block_simage_index(2, C_, C_, dil, (C.c_long * 8)(*[int(v) for v in offs]), idx.ctypes.data_as(C.c_void_p)), "index")
    img = torch.empty(n, device="cuda")
    _lib.check(lib.nsc_gather(pd.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st), "gather")
    x, dy = torch.randn(B, C_, T, device="cuda"), torch.randn(B, C_, T, device="cuda")
    h, lin = torch.randn(B, 20, T, device="cuda"), torch.randn(B, 20, T, device="cuda")
    th = torch.tanh(torch.randn(B, 20, T, device="cuda"))
    dx, da, dz1 = torch.empty_like(x), torch.empty(B, 40, T, device="cuda"), torch.empty(B, 20, T, device="cuda")
    for which in ("k9", "k15"):
        # the stamps are overwritten by every launch: in_act = -9 (probes build) stops after the k9 launch(es)
        def run():
            _lib.check(lib.nsc_gated_block_dgrad_simg2(P(img), P(pd), P(x), P(h), P(lin), P(th), P(dy), None, P(da), P(dz1),
                                                       B, C_, C_, T, dil, 2 if which == "k15" else -9, st), "simg2")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 128)()
        assert lib.nsc_probe_read_bb(buf) == 0
        v, w4 = list(buf)[0:8], list(buf)[64:72]
        print(f"B={B} C={C_} T={T} dil={dil}: {'k9 launch alone' if which == 'k9' else 'k9 + k15 launches'} {1e3 * e0.elapsed_time(e1) / 20:.1f} us; "
              f"stamps of the last launch ({which}), last tile of workgroup 0: wave 0 | wave 4 (shader clocks)")
        for i in range(1, 7):
            print(f"   {names[i]:>10}: +{v[i] - v[i - 1]:7d} | +{w4[i] - w4[i - 1]:7d}")
