"""Only the op-surface step (no engine), a few iterations: for rocprofv3 --kernel-trace --stats.   python3 tools/op_surface_prof.py [fused=1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd import loss_terms_and_measures as L, nn_core_operator as nn
from nsc_amd.neural_speech_coding_module import neuralSpeechCodingModule
from nsc_amd.scope import VariableStore, set_store

from nsc_amd import ops as _ops
_T = {}
if os.environ.get("NSC_NODE_TIMES"):
    def _wrap(cls, which):
        f = getattr(cls, which)
        key = f"{cls.__name__}.{which}"
        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            d = _T.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += time.perf_counter() - t
            return r
        setattr(cls, which, staticmethod(g))
    for _n in dir(_ops):
        _c = getattr(_ops, _n)
        if isinstance(_c, type) and issubclass(_c, torch.autograd.Function) and _c is not torch.autograd.Function:
            _wrap(_c, "forward")
            _wrap(_c, "backward")
nn.FUSED_BLOCKS = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
B = 128
dev = torch.device("cuda", 0)
_, _, x_np, _ = bench.synth_batch(B, 0, dev)
xd = torch.from_numpy(x_np).to(dev).reshape(B, 512, 1)
tgt = xd[:, :, 0].contiguous()
st = VariableStore(device="cuda:0")
set_store(st)
m = neuralSpeechCodingModule.__new__(neuralSpeechCodingModule)
m._bottleneck_kernel_and_dilation = list(bench.BKD)


def step():
    st.begin_pass()
    for v in st.vars.values():
        v.grad = None
    p, _, _, _, decoded, _, _, _ = m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
    loss = (60.0 * L.mse_loss(decoded, tgt) + 10.0 * L.mfcc_loss(decoded, tgt) + 10.0 * L.quan_loss(p)).sum() + B * 0.3 * L.entropy_coding_loss(p)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
_T.clear()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for k, (n, t) in sorted(_T.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:28s} {n / 20:5.1f} calls/step {1e6 * t / 20:8.1f} us/step  {1e6 * t / n:6.1f} us/call")
if _T:
    print(f"   sum of node bodies {1e6 * sum(t for _, t in _T.values()) / 20:.1f} us/step")
print(f"20 steps: host enqueue {1e3 * (t1 - t0) / 20:.3f} ms/step, with final sync {1e3 * (t2 - t0) / 20:.3f} ms/step")
if os.environ.get("NSC_CPROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
