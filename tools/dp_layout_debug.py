"""Root-cause helper for tests/test_engine_gpu.py::test_data_parallel_engine_two_ranks_equals_one_process[True].

Run under torch.distributed.run with 2 ranks sharing one GPU (gloo):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29655 tools/dp_layout_debug.py

Computes the config-3 (LPC) two-rank step in every (layout, launch mode) pair - tail / overlap gradient messages, eager / segmented
hipGraph replay - and prints, for each, the gradient tensor and index with the largest difference from the one-process step at B = 4.
NSC_TAIL_OVERLAP=0 removes the second stream at the tail; NSC_DEBUG_LPC=0 runs the non-LPC step.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nsc_amd.dist import Comm
from nsc_amd.engine import CascadeEngine
from tests._util import BKD, make_store, synth_frames, dev

comm = Comm(backend="gloo")
B, Bl = 4, 2
LPC = os.environ.get("NSC_DEBUG_LPC", "1") == "1"
ps = make_store(2, [[2], [2]], [32, 32], lpc=LPC)
x = synth_frames(B)
cfg = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0, 10.0], c_ent=[0.3, 0.5], trainable=[True, True], lr=2e-4, slot=1)
kw = {}
lpc_all = None
if LPC:
    ps.params["lpc_quan/alpha"] = np.array(-40.0)
    cfg.update(c_quan_lpc=10.0, train_lpc=True, quan_op=True)
    kw = dict(res_scalar=2.0, scale_first=True, lpc=True)
    lpc_all = np.sort(np.random.default_rng(3).uniform(0.03, 3.1, (B, 16, 1)), axis=1).astype(np.float32)
lo, hi = comm.shard(B)
eng = CascadeEngine(Bl, 2, BKD, [[2], [2]], [32, 32], **kw)
xd = dev(x[lo:hi].transpose(0, 2, 1))
lx = dev(lpc_all[lo:hi]) if LPC else None


def named_grads(e):
    torch.cuda.synchronize()
    return {k: v.copy() for k, v in e.named("grads").items()}


runs = {}
for overlap in (False, True):
    eng.dp_overlap = overlap
    lay = "overlap" if overlap else "tail"
    for rep in range(2):                       # twice: the second eager run starts from a used engine, like the replay does
        eng.load_named(ps.params); eng.reset_adam()
        eng.train_step(xd, xd, cfg, lpc_x=lx, comm=comm)
        runs[f"{lay}.eager{rep}"] = named_grads(eng)
    eng.load_named(ps.params); eng.reset_adam()
    st = eng.capture_train_step(xd, xd, cfg, lpc_x=lx, comm=comm)
    for rep in range(2):
        eng.load_named(ps.params); eng.reset_adam()
        st.replay()
        runs[f"{lay}.replay{rep}"] = named_grads(eng)

if comm.rank == 0:
    ref = CascadeEngine(B, 2, BKD, [[2], [2]], [32, 32], **kw)
    ref.load_named(ps.params)
    xf = dev(x.transpose(0, 2, 1))
    ref.train_step(xf, xf, cfg, lpc_x=dev(lpc_all) if LPC else None)
    g2 = named_grads(ref)
    gmax = max(float(np.abs(v).max()) for v in g2.values())
    print(f"tail_overlap={eng.tail_overlap} lpc={LPC} global max |g| = {gmax:.6g} "
          f"at {max(g2, key=lambda k: float(np.abs(g2[k]).max()))}")
    for name, g in runs.items():
        worst = []
        for k, v in g.items():
            d = np.abs(v - g2[k])
            i = int(d.argmax()) if d.size else 0
            worst.append((float(d.max()) if d.size else 0.0, k, i, float(v.reshape(-1)[i]) if d.size else 0.0,
                          float(g2[k].reshape(-1)[i]) if d.size else 0.0, float(np.abs(g2[k]).max()) if d.size else 0.0))
        worst.sort(reverse=True)
        print(f"--- {name}: max diff / global max = {worst[0][0] / gmax:.3e}")
        for dmax, k, i, a, b, tmax in worst[:4]:
            print(f"    {k}[{i}]: got {a:.9g} want {b:.9g} diff {dmax:.3e} ({dmax / gmax:.2e} of global max, "
                  f"{dmax / max(tmax, 1e-30):.2e} of the tensor's max {tmax:.4g})")
comm.barrier()
comm.close()
