"""A/B timing of the gated-block data gradient: the exact fused kernel (nsc_gated_block_dgrad_img) against the three-launch split path
(nsc_gated_block_dgrad_simg2: csrc/block_bwd_split.hip), per shape of the headline step.  The pair launches of the engine run two
exact blocks in ~1.9x the single-block time (profiles/r05k_*): compare with 2x the three-launch figure."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
P = lambda t: t.data_ptr()


def image(split, which, C_, Cin, dil, src, o):
    fn_n = lib.nsc_gated_block_simage_words if split else lib.nsc_gated_block_image_floats
    fn_i = lib.nsc_gated_block_simage_index if split else lib.nsc_gated_block_image_index
    n = int(fn_n(which, C_, Cin, dil))
    idx = np.empty(n, np.int32)
    _lib.check(fn_i(which, C_, Cin, dil, (C.c_long * len(o))(*[int(v) for v in o]), idx.ctypes.data_as(C.c_void_p)), "index")
    img = torch.empty(n, device="cuda")
    _lib.check(lib.nsc_gather(src.data_ptr(), torch.tensor(idx, device="cuda").data_ptr(), img.data_ptr(), n, st), "gather")
    return img


def timeit(run, n=20):
    """n launches replayed from a hipGraph (the engine's step is one): eager launches of 10-us kernels measure the host."""
    global st
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    keep = st
    with torch.cuda.stream(side):
        st = side.cuda_stream
        g.capture_begin()
        for _ in range(n):
            run()
        g.capture_end()
    st = keep
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for (B, C_, T, dil) in [(128, 100, 512, 1), (128, 100, 512, 2), (128, 100, 256, 1), (128, 100, 256, 2), (128, 50, 512, 1), (128, 50, 512, 2),
                        (1024, 100, 256, 2)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    w = [f(1, C_, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
    pd = torch.tensor(np.concatenate([a.reshape(-1) for a in w]), device="cuda")
    wt = [np.ascontiguousarray(w[i][::-1].transpose(0, 2, 1)) for i in (0, 2, 4, 6)]
    td = torch.tensor(np.concatenate([a.reshape(-1) for a in wt]), device="cuda")
    toffs = np.concatenate([[0], np.cumsum([a.size for a in wt])[:-1]]).astype(np.int64)
    x, dy = torch.randn(B, C_, T, device="cuda"), torch.randn(B, C_, T, device="cuda")
    h, lin = torch.randn(B, 20, T, device="cuda"), torch.randn(B, 20, T, device="cuda")
    th = torch.tanh(torch.randn(B, 20, T, device="cuda"))
    dx, da, dz1 = torch.empty_like(x), torch.empty(B, 40, T, device="cuda"), torch.empty(B, 20, T, device="cuda")
    fl = 2.0 * B * T * (C_ * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C_)
    img_e = image(False, 1, C_, C_, dil, td, toffs)
    img_s = image(True, 2, C_, C_, dil, pd, offs)
    t_e = timeit(lambda: _lib.check(lib.nsc_gated_block_dgrad_img(P(img_e), P(x), P(h), P(lin), P(th), P(dy), P(dx), P(da), P(da) + 80 * T, P(dz1),
                                                                   B, C_, C_, T, dil, 2, 40, st), "dgrad"))
    full = lambda dxp: _lib.check(lib.nsc_gated_block_dgrad_simg2(P(img_s), P(pd), P(x), P(h), P(lin), P(th), P(dy), dxp, P(da), P(dz1), B, C_, C_, T,
                                                                   dil, 2, st), "dgrad simg2")
    t_s = timeit(lambda: full(P(dx)))
    t_2 = timeit(lambda: full(None))                     # the two GEMM launches alone
    print(f"B={B:5d} C={C_:3d} T={T} dil={dil}: exact {t_e:7.1f} us {fl / t_e / 1e6:6.1f} TF | three launches {t_s:7.1f} us {fl / t_s / 1e6:6.1f} TF "
          f"(GEMMs {t_2:6.1f}, 1x1 {t_s - t_2:5.1f}) | x{t_e / t_s:.2f}", flush=True)
