"""Stress test behind the intermittent red of test_data_parallel_engine_two_ranks_equals_one_process[True] (round 5 driver run,
reproduced once in 24 steps in round 6): repeat the config-3 data-parallel step many times per (tail stream, message layout) pair
and compare every repetition with the first - gradients AND every engine buffer (checksums), so that a glitch names the first
tensor that differs.

    python tools/dp_race_stress.py [reps]                                        (one process, collectives are no-ops)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29661 tools/dp_race_stress.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nsc_amd.dist import Comm
from nsc_amd.engine import CascadeEngine
from tests._util import BKD, make_store, synth_frames, dev

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SOLO = os.environ.get("NSC_STRESS_SOLO", "") == "1"     # no communicator at all: the plain single-GPU step (run two such processes
                                                       # side by side to see what SHARING the GPU alone does)
comm = Comm(backend="gloo")
B = int(os.environ.get("NSC_STRESS_B", "4"))
Bl = B // comm.world
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
if os.environ.get("NSC_STRESS_PAIRS", "") == "0":
    eng.fused_pairs = False
xd = dev(x[lo:hi].transpose(0, 2, 1))
lx = dev(lpc_all[lo:hi]) if LPC else None


RESTORE = os.environ.get("NSC_STRESS_RESTORE", "h2d")   # how the parameters are put back between repetitions: h2d = load_named (a host->device
                                                        # copy, what the tests do) | d2d = a device-side copy | lr0 = never changed (lr = 0)
if RESTORE == "lr0":
    cfg["lr"] = 0.0
eng.load_named(ps.params)
torch.cuda.synchronize()
saved = eng.params.clone()


def restore():
    if RESTORE == "h2d":
        eng.load_named(ps.params)
    elif RESTORE == "d2d":
        eng.set_params(saved)
    eng.reset_adam()


WATCH = ("scope_1.code", "scope_2.code", "scope_1.dec", "scope_2.dec", "dsum1", "ddec0", "scope_1.b4.out", "scope_1.h0")
                                                       # buffers kept elementwise: the usual first-to-differ ones and their producers' inputs


def snapshot():
    torch.cuda.synchronize()
    g = eng.grads.detach().cpu().numpy().copy()
    sums = {k: float(v.double().sum().item()) for k, v in eng._bufs.items() if v.dtype == torch.float32}
    sums["__watch__"] = {k: eng._bufs[k].detach().cpu().numpy().copy() for k in WATCH if k in eng._bufs}
    return g, sums


def describe(name, a, b):
    """Where and how two copies of a [B, C, T] buffer differ: per (frame, channel) rows, runs along time."""
    d = np.abs(a - b)
    nz = np.argwhere(d > 0)
    if nz.size == 0:
        return f"{name}: identical"
    rows = {}
    for f, c, t in nz:
        rows.setdefault((int(f), int(c)), []).append(int(t))
    parts = []
    for (f, c), ts in list(rows.items())[:6]:
        ts = sorted(ts)
        lst = f" steps {ts} signed diffs {[float(f'{v:.2e}') for v in (a - b)[f, c, ts]]}" if len(ts) <= 32 else ""
        parts.append(f"frame {f} ch {c}: {len(ts)} steps in [{ts[0]}, {ts[-1]}], max |diff| {d[f, c, ts].max():.3e} of |value| <= {np.abs(b[f, c, ts]).max():.3e}{lst}")
    return f"{name}: {len(nz)} elements in {len(rows)} (frame, channel) rows; " + " | ".join(parts)


def name_of(i):
    for name, (off, shape) in eng.layout.entries.items():
        n = int(np.prod(shape)) if len(shape) else 1
        if off <= i < off + n:
            return f"{name}[{i - off}]"
    return str(i)


for tail in (True, False):
    for overlap in (True, False):
        eng.tail_overlap = tail
        eng.dp_overlap = overlap
        base = None
        bad = 0
        worst = 0.0
        for r in range(reps):
            restore()
            eng.train_step(xd, xd, cfg, lpc_x=lx, comm=None if SOLO else comm)
            g, sums = snapshot()
            if base is None:
                base = (g, sums)
                continue
            d = np.abs(g - base[0])
            rel = float(d.max() / np.abs(base[0]).max())
            worst = max(worst, rel)
            if rel > 1e-6:
                bad += 1
                if bad <= 3 and comm.rank == 0:
                    i = int(d.argmax())
                    # buffers in creation order = the order of their first use in a step: the first one named is the origin
                    diff = [(k, f"{abs(sums[k] - base[1][k]) / (abs(base[1][k]) + 1e-30):.1e}") for k in sums if k in base[1] and k != "__watch__"
                            and abs(sums[k] - base[1][k]) > 1e-9 * (abs(base[1][k]) + 1e-30)]
                    for k, v in sums["__watch__"].items():
                        print("    " + describe(k, v, base[1]["__watch__"][k]))
                    print(f"  glitch at rep {r}: rel {rel:.3e} at {name_of(i)}; {int((d > 1e-6 * np.abs(base[0]).max()).sum())} gradient "
                          f"entries off; buffers whose checksum moved, in creation order: {diff[:10]} ... {len(diff)} in all")
        if comm.rank == 0:
            print(f"restore={RESTORE} tail_overlap={tail} dp_overlap={overlap} world={comm.world} fused_pairs={eng.fused_pairs}: {bad} glitches in "
                  f"{reps - 1} repetitions, worst rel diff {worst:.3e}", flush=True)
comm.barrier()
comm.close()
