"""Which torch (aten) operators one op-surface step still runs, with the Python line that asked for each: the surface's own arithmetic is
libnsc_hip.so launches, so every aten kernel is either the user's loss arithmetic, autograd's bookkeeping (gradient clones / sums) or an
allocation helper.   python tools/op_surface_aten.py"""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd import loss_terms_and_measures as L
from nsc_amd.neural_speech_coding_module import neuralSpeechCodingModule
from nsc_amd.scope import VariableStore, set_store

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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
seen = Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::clone", "aten::copy_", "aten::zeros", "aten::fill_", "aten::add", "aten::add_", "aten::mul",
                                                   "aten::sum", "aten::contiguous", "aten::zero_", "aten::ones_like", "aten::expand", "aten::_to_copy"):
        frames = [f for f in (e.stack or []) if "nsc_amd" in f or "op_surface_aten" in f or "autograd" in f]
        seen[(e.name, frames[0] if frames else "(autograd engine / no python frame)")] += 1
for (name, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:18s} {where}")
