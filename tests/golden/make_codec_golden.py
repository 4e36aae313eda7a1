#!/usr/bin/env python3
"""Generate tests/golden/codec_golden.npz from the float64 NumPy oracle (seeded; SURVEY 8c plan).
Inputs + expected outputs only.  Run: python tests/golden/make_codec_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nsc_oracle as O            # noqa: E402
from oracle import nsc_oracle_torch as OT     # noqa: E402
from tests.test_oracle import _setup, BKD     # noqa: E402

ps, x = _setup([2], B=2)
ps.begin_replay()
o = O.codec_forward(x, ps, "scope_1", BKD, [2], 32, 1.0, True)
tgt = x[:, :, 0]
coeff, tau = [60.0, 10.0, 10.0, 0.0], 0.3
tp = OT.TorchParams(ps)
ot = OT.codec_forward(torch.tensor(x), tp, "scope_1", BKD, [2], 1.0, True)
loss = OT.total_loss_sum(ot["decoded"], torch.tensor(tgt), [ot["p"]], coeff, tau, "quan_last")
loss.backward()
flat_grad = np.concatenate([tp.t[k].grad.numpy().ravel() for k in tp.names])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "codec_golden.npz"),
                    x=x, decoded=o["decoded"], floating_code=o["floating_code"], code=o["code"],
                    p_sum_hist=o["p"].reshape(-1, 32).sum(0),
                    time_loss=O.mse_loss(o["decoded"], tgt), freq_loss=O.mfcc_loss(o["decoded"], tgt),
                    quan_loss=O.quan_loss(o["p"]), ent_loss=O.entropy_coding_loss(o["p"]),
                    total_loss=float(loss.detach()), flat_grad=flat_grad.astype(np.float32),
                    coeff=np.array(coeff), tau=np.array(tau))
print("wrote codec_golden.npz; total loss", float(loss), "grad norm", np.linalg.norm(flat_grad))
