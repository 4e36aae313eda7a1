"""Replay of the reference-executed training phases (tests/golden/reference_exec.npz) with the float64 torch oracle.
Shared by the CPU oracle-pinning tests and the GPU parity tests (which take the gradients / trajectories from here
only where the fixture holds a strided sample, and the fixture's numbers themselves where it holds them in full)."""
import json
import os

import numpy as np
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FX = np.load(os.path.join(GOLD, "reference_exec.npz"))
BKD = [9, 9, 100, 20, 1, 2]
STRIDE = 31
COEFF = [60.0, 10.0, 10.0, 0.3]


def sample(v):
    v = np.asarray(v, np.float64).ravel()
    return np.concatenate([[v.sum(), (v * v).sum()], v[::STRIDE]])


def lsf_table():
    kats = json.load(open(os.path.join(GOLD, "reference_kats.json")))
    return np.array(kats["lsf_bins"], np.float32).astype(np.float64)


def named_store(num_codecs, strides, bins, lpc=False):
    """Oracle ParamStore whose kernels are the name-seeded values the reference run started from."""
    ps = O.ParamStore(name_seeded=True)
    if lpc:
        ps.var("lpc_quan", "alpha", O.INIT_ALPHA)
        ps.var("lpc_quan", "bins", lsf_table())
    x0 = np.zeros((1, 512, 1))
    for i in range(num_codecs):
        O.codec_forward(x0, ps, f"scope_{i + 1}", BKD, strides[i], bins[i], 0.0, True)
    return ps


def _close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.max(np.abs(b))), 1e-30)
    err = float(np.max(np.abs(a - b))) / scale
    assert err <= tol, f"{what}: max err {err:.3e} of the tensor's max"


class PhaseReplay:
    """Walks the recorded steps of one flow ('td' or 'lp') with the torch oracle and checks, against the numbers the
    reference's code produced: each step's loss vector, the first gradient of each optimizer, the checkpoint."""

    def __init__(self, flow, lpc, res_scalar=1.0, tol=1e-9):
        self.flow, self.lpc, self.rs, self.tol = flow, lpc, res_scalar, tol
        self.ps = named_store(2, [[2], [2]], [32, 32], lpc=lpc)
        self.init = {k: np.array(v, np.float64) for k, v in self.ps.params.items()}
        self.tp = OT.TorchParams(self.ps)
        self.checked = dict(loss=0, grad_sets=0, ckpt=0)

    def reinit(self, scopes):
        """variables_initializer(new scope) at the start of a follower phase (cmrl.py:116-119)."""
        for k, v in self.init.items():
            if any(k.startswith(s + "/") for s in scopes):
                self.tp.t[k] = torch.tensor(v, dtype=torch.float64, requires_grad=True)

    def as_store(self):
        ps = O.ParamStore()
        for k in self.ps.params:
            ps.params[k] = self.tp.t[k].detach().numpy().copy()
        ps.begin_replay()
        return ps

    def forward(self, s, pre, num, qon):
        """Returns (loss inputs) for recorded step s."""
        if self.lpc:
            x = torch.tensor(FX[pre + "res_x"][s])
            lx = torch.tensor(FX[pre + "lpc_x"][s])
            pl, _ = OT.scalar_softmax_quantization(lx, self.tp.t["lpc_quan/alpha"], self.tp.t["lpc_quan/bins"], qon, True)
            outs, dec = OT.cascade_forward(x, self.tp, BKD, [[2]] * num, qon, True, res_scalar=self.rs, scale_first=True)
            return x, outs, dec, (pl,)
        x = torch.tensor(FX[pre + "x"][s])
        outs, dec = OT.cascade_forward(x, self.tp, BKD, [[2]] * num, qon, True, res_scalar=self.rs)
        return x, outs, dec, ()

    def run(self, phase, num, mode, train, check=True, after_step=None, on_step=None):
        pre = f"{self.flow}_{phase}_"
        names = [k for k in self.ps.params if any(k.startswith(s + "/") for s in train)]
        adam = {}
        n = len(FX[pre + "loss"])
        for s in range(n):
            qon, lr, opt = float(FX[pre + "qon"][s]), float(FX[pre + "lr"][s]), int(FX[pre + "opt"][s])
            tau = FX[pre + "tau"][s]
            x, outs, dec, p_extra = self.forward(s, pre, num, qon)
            m = "no_quan" if opt == 0 else mode
            lv = OT.phase_loss(dec, x[:, :, 0], [o["p"] for o in outs], COEFF, tau, m, p_extra)
            params = [self.tp.t[k] for k in names]
            grads = torch.autograd.grad(lv.sum(), params, allow_unused=True)
            if check:
                assert int(FX[pre + "nvars"][s]) == len(names)
                _close(lv.detach().numpy(), FX[pre + "loss"][s], self.tol, f"{pre} step {s} loss vector")
                self.checked["loss"] += 1
            st = adam.setdefault(opt, dict(t=0, m={}, v={}))
            if st["t"] == 0 and check:
                for k, g in zip(names, grads):
                    key = f"{pre}grad{opt}|{k}"
                    if g is None:
                        assert key not in FX.files, key
                        continue
                    want = FX[key]
                    scale = max(float(np.sqrt(want[1] / max(g.numel(), 1))), 1e-30)   # rms of the reference gradient
                    got = sample(g.numpy())
                    assert np.max(np.abs(got[2:] - want[2:])) <= 1e-8 * scale + 1e-14, key
                    assert abs(got[1] - want[1]) <= 1e-8 * max(want[1], 1e-30) + 1e-24, key
                self.checked["grad_sets"] += 1
            if on_step is not None:
                on_step(s, dict(x=x, outs=outs, dec=dec, loss=lv, grads=dict(zip(names, grads)), opt=opt, mode=m,
                                qon=qon, tau=tau, lr=lr))
            st["t"] += 1
            t = st["t"]
            lr_t = lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
            with torch.no_grad():
                for k, g in zip(names, grads):
                    if g is None:
                        continue
                    mm = st["m"].get(k, torch.zeros_like(g))
                    vv = st["v"].get(k, torch.zeros_like(g))
                    mm = 0.9 * mm + 0.1 * g
                    vv = 0.999 * vv + 0.001 * g * g
                    st["m"][k], st["v"][k] = mm, vv
                    self.tp.t[k] = (self.tp.t[k] - lr_t * mm / (vv.sqrt() + 1e-8)).detach().requires_grad_(True)
            if after_step is not None:
                after_step(s)
        if check:
            for k in FX.files:
                if k.startswith(pre + "ckpt|"):
                    name = k.split("|", 1)[1]
                    if name in self.tp.t:
                        got, want = sample(self.tp.t[name].detach().numpy()), FX[k]
                        assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, float(np.max(np.abs(want)))), k
            self.checked["ckpt"] += 1
