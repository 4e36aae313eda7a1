"""How many leaky-relu kinks does the split-operand arithmetic flip against the exact fp32 kernels, and at which magnitudes?
Headline configuration (2-codec LPC cascade, B = 128, bench's synthetic batch): one forward per arm on the same parameters; for every
gated block the saved h (lrelu of the 1x1) and the block output are compared elementwise: a FLIP = the two arms disagree on the sign.
A flip at |value| ~ 1e-8 of the tensor's scale is rounding noise on either side of the kink; a flip at a larger magnitude would be an error."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd.engine import CascadeEngine

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "128"))
x, lpc, _, _ = bench.synth_batch(B, 0, dev)
res = {}
for arm in ("exact", "split"):
    eng = CascadeEngine(B, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True, device=dev)
    eng.split_fwd = arm == "split"
    eng.refresh_wt()
    dec = eng.forward(x, 1.0, True, lpc_x=lpc).clone()
    torch.cuda.synchronize()
    t = {"decoded": dec}
    for ci, c in enumerate(eng.codecs):
        for bi, blk in enumerate(c.all_blocks()):
            t[f"codec{ci} block{bi} h"] = blk.h.clone()
            t[f"codec{ci} block{bi} out"] = blk.out.clone()
    res[arm] = t
tot = 0
for k in res["exact"]:
    a, b = res["exact"][k].double(), res["split"][k].double()
    rms = float(a.pow(2).mean().sqrt())
    d = float((a - b).abs().max())
    flips = (torch.sign(a) != torch.sign(b)) & ((a != 0) | (b != 0))
    nf = int(flips.sum())
    tot += nf
    mag = float(torch.maximum(a.abs(), b.abs())[flips].max()) if nf else 0.0
    print(f"{k:22s}: rms {rms:.3e}  max |exact - split| {d:.2e} ({d / rms:.1e} of rms)  sign flips {nf} of {a.numel()}, largest |value| at a flip {mag:.2e}")
print("total flips", tot)
