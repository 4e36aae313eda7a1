import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nsc_oracle_torch as OT
from tests._util import BKD, dev, make_store, synth_frames, relerr
from nsc_amd.engine import CascadeEngine

def run(rs, scale_first, lpc, cq_lpc):
    B = 2
    ps = make_store(2, [[2], [2]], [32, 32], lpc=lpc)
    x = synth_frames(B)
    rng = np.random.default_rng(3)
    lpc_x = np.sort(rng.uniform(0.03, 3.1, (B, 16, 1)), axis=1).astype(np.float32).astype(np.float64)
    coeff = [60.0, 10.0, 10.0, 0.0]
    tp = OT.TorchParams(ps)
    xt = torch.tensor(x)
    outs, dec = OT.cascade_forward(xt, tp, BKD, [[2], [2]], 1.0, True, rs, scale_first)
    for o in outs:
        for k in ("decoded", "code", "floating_code"):
            o[k].retain_grad()
    dec.retain_grad()
    pe = ()
    if lpc and cq_lpc:
        pe = (OT.scalar_softmax_quantization(torch.tensor(lpc_x), tp.t["lpc_quan/alpha"], tp.t["lpc_quan/bins"], 1.0, True)[0],)
    OT.total_loss_sum(dec, xt[:, :, 0], [o["p"] for o in outs], coeff, 0.0, "finetune_lpc", pe).backward()
    eng = CascadeEngine(B, 2, BKD, [[2], [2]], [32, 32], res_scalar=rs, scale_first=scale_first, lpc=lpc)
    eng.load_named(ps.params); eng.refresh_wt()
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    eng.forward(xd, 1.0, True, lpc_x=dev(lpc_x) if lpc else None)
    eng.loss_backward(xd, coeff[0], coeff[1], [coeff[2]] * 2, [0.0, 0.0], [True, True], c_quan_lpc=cq_lpc)
    torch.cuda.synchronize()
    print(f"--- rs={rs} scale_first={scale_first} lpc={lpc} cq_lpc={cq_lpc}")
    print("decoded", relerr(eng.decoded.cpu().numpy()[:, 0], dec.detach().numpy()))
    print("G", relerr(eng._bufs["loss.G"].cpu().numpy()[:, 0], dec.grad.numpy()))
    for i in range(2):
        c = eng.codecs[i]
        print(i, "dec", relerr(c.dec.cpu().numpy()[:, 0], outs[i]["decoded"].detach().numpy()),
              "ddec", relerr(eng._bufs[f"ddec{i}"].cpu().numpy()[:, 0], outs[i]["decoded"].grad.numpy()),
              "qcode", relerr(c.qcode.cpu().numpy()[:, 0], outs[i]["code"].detach().numpy()[:, :, 0]),
              "dq", relerr(eng._bufs[f"scope_{i+1}.b5.dx"].cpu().numpy()[:, 0], outs[i]["code"].grad.numpy()[:, :, 0]))
    mine = eng.named("grads")
    for k in tp.names:
        g = tp.t[k].grad
        if g is None: continue
        e = relerr(mine[k], g.numpy())
        if e > 3e-4: print("  BAD", k, f"{e:.2e}")

run(2.0, True, True, 10.0)
run(2.0, True, False, 0.0)
run(2.0, False, False, 0.0)
run(1.0, True, False, 0.0)

print("==== sign flip check")
B = 2
ps = make_store(2, [[2], [2]], [32, 32], lpc=False)
x = synth_frames(B)
eng = CascadeEngine(B, 2, BKD, [[2], [2]], [32, 32], res_scalar=2.0, scale_first=True)
eng.load_named(ps.params); eng.refresh_wt()
xd = dev(x.transpose(0, 2, 1))
eng.forward(xd, 1.0, True)
torch.cuda.synchronize()
c = eng.codecs[1]
xin, dwo, up = c.up_saved[0]
W = ps.params["scope_2/separable_conv1d/pointwise_kernel"][0]   # [C, Cout]
bias = ps.params["scope_2/separable_conv1d/bias"]
z = np.einsum("bct,co->bot", dwo.cpu().numpy().astype(np.float64), W) + bias[None, :, None]   # [B,100,256]
upn = up.cpu().numpy()          # [B,50,512]
unsh = np.empty_like(z)
for ch in range(100):
    unsh[:, ch, :] = upn[:, ch >> 1, (ch & 1)::2]
print("min |z|", np.abs(z).min(), "sign mismatches", int(((z > 0) != (unsh > 0)).sum()), "of", z.size)
idx = np.argsort(np.abs(z).ravel())[:5]
print("smallest |z| f64:", z.ravel()[idx], "engine:", unsh.ravel()[idx])
