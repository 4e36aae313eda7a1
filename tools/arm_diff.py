"""Exact-fp32 arm against the split-operand arm on the same parameters and inputs: per parameter tensor, the largest gradient
difference relative to the tensor's largest gradient - to localise a disagreement to a layer.   B=2 FOLLOWER=1 python tools/arm_diff.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd.engine import CascadeEngine

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "2"))
follower = os.environ.get("FOLLOWER", "1") == "1"
x, lpc, _, _ = bench.synth_batch(max(B, 2), 0, dev)
x, lpc = x[:B].contiguous(), lpc[:B].contiguous()
res = {}
for arm in os.environ.get("ARMS", "exact,fwd,wgrad,dgrad").split(","):
    eng = CascadeEngine(B, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True, device=dev)
    eng.split_fwd = arm == "fwd"
    eng.split_wgrad_arith = arm == "wgrad"
    eng.split_dgrad = arm == "dgrad"
    eng.refresh_wt()
    eng.grads.zero_()
    eng.forward(x, 1.0, True, lpc_x=lpc)
    tr = [False, True] if follower else [True, True]
    eng.loss_backward(x, 60.0, 10.0, [10.0, 10.0], [0.0, 0.0], tr, c_quan_lpc=10.0)
    torch.cuda.synchronize()
    res[arm] = eng.named("grads")
base = res["exact"]
for arm in res:
    if arm == "exact":
        continue
    rows = []
    for k, g in base.items():
        m = float(np.abs(g).max())
        if m == 0:
            continue
        rows.append((float(np.abs(res[arm][k] - g).max()) / m, k, m))
    rows.sort(reverse=True)
    print(f"--- {arm} (only this kernel family split) vs exact: largest |dgrad| / max |grad| per tensor")
    for r, k, m in rows[:12]:
        print(f"   {r:.2e}  {k}  (max |grad| {m:.2e})")
