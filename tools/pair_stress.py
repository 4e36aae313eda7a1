"""Soak test of the pair launches (nsc_gated_block_pair_fwd_img / _dgrad_img): N repetitions per shape against the two-launch result,
bit for bit, with a second stream keeping the GPU busy with unrelated kernels (different timing every repetition); the launches'
time-out counter must stay 0.  usage: python tools/pair_stress.py [reps]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
st = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
rng = np.random.default_rng(0)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
nfl = int(lib.nsc_gated_block_pair_flag_ints())
P = lambda t: t.data_ptr()
bad = 0
for (C_, T, B, Cin0) in [(100, 256, 128, 100), (100, 512, 128, 100), (50, 512, 128, 50), (100, 256, 128, 1), (25, 128, 256, 25)]:
    f = lambda *sh: (0.1 * rng.standard_normal(sh)).astype(np.float32)
    imgs = []
    for dil in (1, 2):
        Ci = Cin0 if dil == 1 else C_
        w = [f(1, Ci, 20), f(20), f(15, 20, 20), f(20), f(15, 20, 20), f(20), f(9, 20, C_), f(C_)]
        flat = np.concatenate([a.reshape(-1) for a in w]); offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64)
        wt = [np.ascontiguousarray(w[i][::-1].transpose(0, 2, 1)) for i in (0, 2, 4, 6)]
        tflat = np.concatenate([a.reshape(-1) for a in wt]); toffs = np.concatenate([[0], np.cumsum([a.size for a in wt])[:-1]]).astype(np.int64)
        pair = []
        for which, src, o in ((0, dev(flat), offs), (1, dev(tflat), toffs)):
            n = int(lib.nsc_gated_block_image_floats(which, C_, Ci, dil)); idx = np.empty(n, np.int32)
            assert lib.nsc_gated_block_image_index(which, C_, Ci, dil, (C.c_long * len(o))(*[int(v) for v in o]), idx.ctypes.data_as(C.c_void_p)) == 0
            img = torch.empty(n, device="cuda"); idt = torch.tensor(idx, device="cuda")
            assert lib.nsc_gather(src.data_ptr(), idt.data_ptr(), img.data_ptr(), n, st) == 0
            torch.cuda.synchronize(); pair.append(img)
        imgs.append(pair)
    (f0, b0), (f1, b1) = imgs
    x = dev(rng.standard_normal((B, Cin0, T))); mk = lambda *sh: torch.empty(sh, device="cuda")
    o0, o1 = mk(B, C_, T), mk(B, C_, T); s0, s1 = [mk(B, 20, T) for _ in range(4)], [mk(B, 20, T) for _ in range(4)]
    assert lib.nsc_gated_block_fwd_img(P(f0), P(x), P(o0), *[P(t) for t in s0], B, C_, Cin0, T, 1, 0, st) == 0
    assert lib.nsc_gated_block_fwd_img(P(f1), P(o0), P(o1), *[P(t) for t in s1], B, C_, C_, T, 2, 1, st) == 0
    dy = dev(rng.standard_normal((B, C_, T))); h0, l0, h1, l1 = (dev(rng.standard_normal((B, 20, T))) for _ in range(4))
    t0_, t1_ = (torch.tanh(dev(rng.standard_normal((B, 20, T)))) for _ in range(2)); x1 = dev(rng.standard_normal((B, C_, T)))
    dx1, da1, dz1 = mk(B, C_, T), mk(B, 40, T), mk(B, 20, T); dx0, da0, dz0 = mk(B, Cin0, T), mk(B, 40, T), mk(B, 20, T)
    act0 = 0 if Cin0 == 1 else 2
    assert lib.nsc_gated_block_dgrad_img(P(b1), P(x1), P(h1), P(l1), P(t1_), P(dy), P(dx1), P(da1), P(da1) + 80 * T, P(dz1), B, C_, C_, T, 2, 2, 40, st) == 0
    assert lib.nsc_gated_block_dgrad_img(P(b0), None if Cin0 == 1 else P(x), P(h0), P(l0), P(t0_), P(dx1), P(dx0), P(da0), P(da0) + 80 * T, P(dz0), B, C_, Cin0, T, 1, act0, 40, st) == 0
    torch.cuda.synchronize()
    p0, p1 = mk(B, C_, T), mk(B, C_, T); q0, q1 = [mk(B, 20, T) for _ in range(4)], [mk(B, 20, T) for _ in range(4)]
    e1, a1, z1 = mk(B, C_, T), mk(B, 40, T), mk(B, 20, T); e0, a0, z0 = mk(B, Cin0, T), mk(B, 40, T), mk(B, 20, T)
    flags = torch.zeros(2 * nfl, dtype=torch.int32, device="cuda")
    tmo = torch.zeros(4, dtype=torch.int32, device="cuda")
    junk = torch.randn(4096, 4096, device="cuda")
    nbad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):                    # unrelated work of varying length on another stream
            for _ in range(rep % 3):
                junk.mul_(1.0001)
        for t in [p0, p1, e1, a1, z1, e0, a0, z0] + q0 + q1:
            t.fill_(float("nan"))
        flags.zero_()
        tmo.zero_()
        assert lib.nsc_gated_block_pair_fwd_img(P(f0), P(f1), P(x), P(p0), *[P(t) for t in q0], P(p1), *[P(t) for t in q1], B, C_, Cin0, T, 1, P(flags), P(tmo), st) == 0
        assert lib.nsc_gated_block_pair_dgrad_img(P(b1), P(x1), P(h1), P(l1), P(t1_), P(dy), P(e1), P(a1), P(z1), P(b0), None if Cin0 == 1 else P(x), P(h0), P(l0),
                                                  P(t0_), P(e0), P(a0), P(z0), B, C_, Cin0, T, act0, P(flags) + 4 * nfl, P(tmo), st) == 0
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip([o0, o1] + s0 + s1 + [dx1, da1, dz1, dx0, da0, dz0], [p0, p1] + q0 + q1 + [e1, a1, z1, e0, a0, z0]))
        to = int(tmo[0])
        if not ok or to:
            nbad += 1
    print(f"C={C_} T={T} B={B} Cin0={Cin0}: {reps} repetitions, {nbad} with a mismatch or a time-out")
    bad += nbad
print("PASS" if bad == 0 else "FAIL")
sys.exit(1 if bad else 0)
