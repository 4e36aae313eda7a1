"""A/B of the wave-specialised block kernels (v3, csrc/block3.hip) against v2 (csrc/block.hip) in ONE process, through
the probes library (`make -C nsc_amd/csrc probes`): max |v3 - v2| of every output, then interleaved timing rounds
(median / min), forward and data-gradient kernels, on the codec's block shapes.

    python tools/block3_ab.py [fwd|dgrad|both]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsc_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "nsc_amd", "libnsc_hip_probes.so")
lib = _lib.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None
which = sys.argv[1] if len(sys.argv) > 1 else "both"


def set_v2(name, on):
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)


def timeit(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def ab(run, envname, outs, label, flops, rounds=7):
    res = {}
    for v2 in (True, False):
        set_v2(envname, v2)
        for o in outs:
            o.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        res[v2] = [o.clone() for o in outs]
    diffs = []
    for a, b in zip(res[True], res[False]):
        bad = torch.isnan(b).sum().item()
        diffs.append(f"{(a - b).abs().max().item():.2e}" + (f" NaN x{bad}" if bad else ""))
    ts = {True: [], False: []}
    for _ in range(rounds):
        for v2 in (True, False):
            set_v2(envname, v2)
            run(); run()
            ts[v2].append(timeit(run))
    set_v2(envname, False)
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    mn = {k: min(v) for k, v in ts.items()}
    print(f"{label}: max|v3-v2| {diffs} | v2 {med[True]:7.1f} us (min {mn[True]:7.1f}, {flops / med[True] / 1e6:5.1f} TF/s)"
          f" | v3 {med[False]:7.1f} us (min {mn[False]:7.1f}, {flops / med[False] / 1e6:5.1f} TF/s) | x{med[True] / med[False]:.3f}",
          flush=True)


SHAPES = [(128, 100, 512, 1, 100), (128, 100, 512, 2, 100), (128, 100, 256, 1, 100), (128, 100, 256, 2, 100),
          (128, 50, 512, 1, 50), (128, 50, 512, 2, 50), (128, 100, 256, 1, 1), (128, 100, 256, 2, 1),
          (4096, 100, 256, 2, 100), (1, 100, 512, 2, 100), (3, 100, 200, 1, 100), (2, 50, 130, 2, 1)]

if which in ("fwd", "both"):
    for (B, C, T, dil, Cin) in SHAPES:
        torch.manual_seed(0)
        x = torch.randn(B, Cin, T, device=dev)
        w1 = torch.randn(1, Cin, 20, device=dev) * 0.1; b1 = torch.randn(20, device=dev) * 0.1
        wl = torch.randn(15, 20, 20, device=dev) * 0.05; wr = torch.randn(15, 20, 20, device=dev) * 0.05
        bl = torch.randn(20, device=dev) * 0.1; br = torch.randn(20, device=dev) * 0.1
        w9 = torch.randn(9, 20, C, device=dev) * 0.05; b9 = torch.randn(C, device=dev) * 0.1
        out = torch.empty(B, C, T, device=dev)
        save = B <= 128
        sv = [torch.empty(B, 20, T, device=dev) for _ in range(4)] if save else [None] * 4
        fn = lib.nsc_gated_block_fwd_cin1 if Cin == 1 else lib.nsc_gated_block_fwd

        def run():
            _lib.check(fn(p(x), p(w1), p(b1), p(wl), p(bl), p(wr), p(br), p(w9), p(b9), p(out), p(sv[0]), p(sv[1]), p(sv[2]),
                          p(sv[3]), B, C, T, 20, 9, dil, 0, st), "blk")
        fl = 2.0 * B * T * (Cin * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C)
        ab(run, "NSC_BLOCK_FWD_V2", [out] + [s for s in sv if s is not None], f"fwd   B={B} C={C} T={T} dil={dil} Cin={Cin}", fl)

if which in ("dgrad", "both"):
    for (B, C, T, dil, Cin) in SHAPES:
        torch.manual_seed(0)
        x = torch.randn(B, Cin, T, device=dev); dy = torch.randn(B, C, T, device=dev)
        h, lin, th = (torch.randn(B, 20, T, device=dev) for _ in range(3)); th = torch.tanh(th)
        wt1 = torch.randn(1, 20, Cin, device=dev) * 0.1
        wtl = torch.randn(15, 20, 20, device=dev) * 0.05; wtr = torch.randn(15, 20, 20, device=dev) * 0.05
        wt9 = torch.randn(9, C, 20, device=dev) * 0.05
        dx = torch.empty_like(x); da = torch.empty(B, 40, T, device=dev); dz1 = torch.empty(B, 20, T, device=dev)
        if Cin == 1:
            def run():
                _lib.check(lib.nsc_gated_block_dgrad_cin1(p(h), p(lin), p(th), p(dy), p(wt1), p(wtl), p(wtr), p(wt9), p(dx),
                                                          p(da[:, :20]), p(da[:, 20:]), p(dz1), B, C, T, 20, 9, dil, 40, st), "dgrad1")
        else:
            def run():
                _lib.check(lib.nsc_gated_block_dgrad(p(x), p(h), p(lin), p(th), p(dy), p(wt1), p(wtl), p(wtr), p(wt9), p(dx), p(da),
                                                     p(dz1), B, C, T, 20, 9, dil, 2, st), "dgrad")
        fl = 2.0 * B * T * (Cin * 20 + 2 * 15 * 20 * 20 + 9 * 20 * C)
        ab(run, "NSC_BLOCK_DGRAD_V2", [dx, da, dz1], f"dgrad B={B} C={C} T={T} dil={dil} Cin={Cin}", fl)
