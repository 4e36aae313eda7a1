"""Debug aid: replay tests/test_reference_exec_gpu.py::test_lpc_phases_on_gpu_match_reference with every engine paired with an
exact-arithmetic twin; after each checked step print where the two arms' block buffers and gradients differ most."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NSC_BLOCK_ARITH", "split")
os.environ.setdefault("NSC_SPLIT_DGRAD", "0")
import tests.test_reference_exec_gpu as M

orig_engine = M._engine


def paired(B, N, strides, bins, ps, **kw):
    eng = orig_engine(B, N, strides, bins, ps, **kw)
    twin = orig_engine(B, N, strides, bins, ps, **kw)
    twin.split_fwd = twin.split_wgrad_arith = twin.split_dgrad = False
    eng.split_wgrad_arith = os.environ.get("DBG_WGRAD", "1") == "1"
    rec = {}
    f0, l0 = eng.forward, eng.loss_backward

    def fwd(*a, **k):
        rec["f"] = (a, k)
        return f0(*a, **k)

    def lb(*a, **k):
        out = l0(*a, **k)
        torch.cuda.synchronize()
        twin.grads.zero_()
        twin.forward(*rec["f"][0], **rec["f"][1])
        twin.loss_backward(*a, **k)
        torch.cuda.synchronize()
        rows = []
        for ci, (c, ct) in enumerate(zip(eng.codecs, twin.codecs)):
            for bi, (b, bt) in enumerate(zip(c.all_blocks(), ct.all_blocks())):
                for nm in ("x", "h", "lin", "th", "g", "out"):
                    u, v = getattr(b, nm, None), getattr(bt, nm, None)
                    if u is None or v is None:
                        continue
                    d = (u.double() - v.double()).abs()
                    rms = float(v.double().pow(2).mean().sqrt()) + 1e-30
                    i = int(d.argmax())
                    rows.append((float(d.max()) / rms, f"codec{ci} block{bi} (C {b.wide} Cin {b.Cin} T {b.T} dil {b.cl.dil}) {nm}", np.unravel_index(i, tuple(u.shape))))
        rows.sort(key=lambda r: -r[0])
        print("  largest block-buffer differences (max |split - exact| / rms):")
        for r in rows[:8]:
            print(f"    {r[0]:.2e}  {r[1]}  at {tuple(int(q) for q in r[2])}")
        g, gt = eng.named("grads"), twin.named("grads")
        rows = sorted(((float(np.abs(g[k] - gt[k]).max()) / (float(np.abs(gt[k]).max()) + 1e-30), k) for k in g if np.abs(gt[k]).max() > 0), reverse=True)
        print("  largest gradient differences (max |split - exact| / max |exact|):", [(f"{r:.1e}", k) for r, k in rows[:8]])
        return out
    eng.forward, eng.loss_backward = fwd, lb
    return eng


M._engine = paired
try:
    M.test_lpc_phases_on_gpu_match_reference()
    print("PASSED")
except AssertionError as e:
    print("FAILED:", str(e)[:600])
