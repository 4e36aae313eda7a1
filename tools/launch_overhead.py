#!/usr/bin/env python3
"""What a dependent kernel launch costs on this stack: N back-to-back launches of a one-workgroup kernel (nsc_increment) and of a
256-workgroup streaming kernel on a small buffer (nsc_axpby, 64 K floats), eager and replayed from a hipGraph.  The per-launch time of
a chain of trivial kernels is the floor every dispatch of a step pays (79 dispatches per step in round 3)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
a = torch.randn(65536, device=dev); b = torch.empty_like(a)
N = 400


def chain(kind, st):
    for _ in range(N):
        if kind == "increment":
            lib.nsc_increment(cnt.data_ptr(), st)
        else:
            lib.nsc_axpby(a.data_ptr(), None, b.data_ptr(), 1.0, 0.0, a.numel(), st)


for kind in ("increment", "axpby64k"):
    st = torch.cuda.current_stream().cuda_stream
    chain(kind, st); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(kind, st); torch.cuda.synchronize(); te = time.perf_counter() - t0
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(kind, s.cuda_stream)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain(kind, torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
    print(f"{kind}: eager {1e6 * te / N:.2f} us per launch, hipGraph replay {1e6 * tg / N:.2f} us per launch ({N} dependent launches)")
