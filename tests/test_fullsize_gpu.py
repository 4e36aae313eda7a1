"""VALUES and size-independent properties at BASELINE.json's full size (config 3: 2-codec CMRL on the fed LPC residual + LSF quantizer,
B = 128 per GPU), where the float64 oracle is too slow to be the checker: determinism of the forward, independence of a
frame's result from the batch it is computed in, linearity of the backward in the loss coefficients, idempotence of the
hard quantizer, and additivity of the gradient over a split of the batch (what the data-parallel SUM all-reduce relies on)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 128


def _setup(b=B):
    import bench
    from nsc_amd.engine import CascadeEngine
    dev = torch.device("cuda", 0)
    eng = CascadeEngine(b, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True, device=dev)
    x, lpc, _, _ = bench.synth_batch(B, 0, dev)
    return bench, eng, x, lpc


def test_forward_is_deterministic_and_batch_independent():
    bench, eng, x, lpc = _setup()
    d1 = eng.forward(x, 1.0, True, lpc_x=lpc).clone()
    d2 = eng.forward(x, 1.0, True, lpc_x=lpc).clone()
    assert torch.equal(d1, d2)                                    # bit-identical run to run
    assert bool(torch.isfinite(d1).all())
    # the same frames through a 64-frame engine: every frame's arithmetic is its own (no cross-frame term in the forward)
    from nsc_amd.engine import CascadeEngine
    e64 = CascadeEngine(64, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True,
                        device=x.device)
    e64.set_params(eng.params)
    dh = e64.forward(x[:64].contiguous(), 1.0, True, lpc_x=lpc[:64].contiguous())
    assert torch.equal(dh, d1[:64])


def test_backward_is_linear_in_the_loss_coefficients():
    bench, eng, x, lpc = _setup()
    eng.refresh_wt()

    def grads(scale):
        eng.grads.zero_()
        eng.forward(x, 1.0, True, lpc_x=lpc)
        eng.loss_backward(x, scale * 60.0, scale * 10.0, [scale * 10.0] * 2, [0.0, 0.0], [True, True], c_quan_lpc=scale * 10.0)
        torch.cuda.synchronize()
        return eng.grads.clone()

    g1, g2 = grads(1.0), grads(2.0)
    scale = float(g1.abs().max())
    assert scale > 0 and bool(torch.isfinite(g1).all())
    assert float((g2 - 2.0 * g1).abs().max()) <= 2e-5 * scale     # fp32 reduction order only (slabs / atomics)


def test_gradient_is_additive_over_a_split_of_the_batch():
    """grad(B = 128) == grad(frames 0..63) + grad(frames 64..127) when no batch-global (entropy) term is on: the identity
    behind the data-parallel SUM all-reduce, at the full per-GPU batch."""
    bench, eng, x, lpc = _setup()
    from nsc_amd.engine import CascadeEngine
    eng.refresh_wt()
    eng.grads.zero_()
    eng.forward(x, 1.0, True, lpc_x=lpc)
    eng.loss_backward(x, 60.0, 10.0, [10.0, 10.0], [0.0, 0.0], [True, True], c_quan_lpc=10.0)
    torch.cuda.synchronize()
    full = eng.grads.clone()
    e64 = CascadeEngine(64, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True,
                        device=x.device)
    e64.set_params(eng.params)
    e64.refresh_wt()
    acc = torch.zeros_like(full)
    for lo in (0, 64):
        xs, ls = x[lo:lo + 64].contiguous(), lpc[lo:lo + 64].contiguous()
        e64.grads.zero_()
        e64.forward(xs, 1.0, True, lpc_x=ls)
        e64.loss_backward(xs, 60.0, 10.0, [10.0, 10.0], [0.0, 0.0], [True, True], c_quan_lpc=10.0)
        torch.cuda.synchronize()
        acc += e64.grads
    scale = float(full.abs().max())
    assert float((acc - full).abs().max()) <= 5e-5 * scale


def test_hard_quantizer_is_idempotent_at_full_size():
    """Quantising already-quantised codes returns them unchanged (bit-exact), 128 x 256 codes on the 32 trained-shape bins."""
    import ctypes as C
    from nsc_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    L, nb = 256, 32
    code = torch.tensor(np.tanh(rng.standard_normal((B, L, 1))).astype(np.float32), device=dev)
    alpha = torch.tensor([-300.0], device=dev)
    bins = torch.linspace(-1, 1, nb, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    q1, q2 = torch.empty_like(code), torch.empty_like(code)
    quan = torch.empty(B, device=dev); hist = torch.zeros(nb, device=dev)
    for src, dst in ((code, q1), (q1, q2)):
        assert lib.nsc_quantize_fwd(src.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, 0, B, L, nb, None, dst.data_ptr(),
                                    quan.data_ptr(), hist.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)
    assert bool(torch.isin(q1.flatten(), bins).all())             # every output is exactly one of the bins
    # alpha = -300: the soft-to-hard quantizer IS nearest-bin rounding (no code is further than half a bin from its output)
    assert float((q1 - code).abs().max()) <= 0.5 * float(bins[1] - bins[0]) + 1e-6


# ---- achieved gradient error per tensor class: printed, written to gpurun_out/ and held to a recorded ceiling ----
# ---- loss terms and forward tensors at B = 128: achieved error recorded and held to 4x what was measured (a regression
# of the mel loss from 2e-5 to 2.9e-4 must not hide under its 3e-4 bound).  Measured on MI355X, round 5 (both arithmetic arms of the
# gated-block kernels: the larger of the two; profiles/r05_loss_term_errors_*.json): max |got - want| / rms(want).
LOSS_CEILING = {          # 4x the larger of the two arms, floor 2e-6 (round 5: profiles/r05_numerics_gate.txt)
    "joint": {"decoded": 8.8e-6, "time loss per frame": 2e-6, "mel loss per frame": 2e-6, "quan loss of codec 1 per frame": 2e-6,
              "quan loss of codec 2 per frame": 2e-6, "LSF quan loss per frame": 2e-6},
    "alpha300": {"decoded": 7.8e-6, "quan loss of codec 1": 2e-6, "quan loss of codec 2": 2e-6, "time loss per frame": 2e-6,
                 "mel loss per frame": 2.2e-6, "LSF quan loss per frame": 2e-6},
}
_LOSS_REC = {}


def _term(test, name, a, b, **kw):
    """assert_close + record: prints the achieved error, the share of elements that needed the rms floor, and holds the achieved
    error to the recorded ceiling (4x measured, floor 2e-6) when one is on record."""
    import json, os
    from tests._util import assert_close
    r = assert_close(a, b, what=name, **kw)
    _LOSS_REC.setdefault(test, {})[name] = r
    print(f"  [{test}] {name}: max err / rms {r['max_err_over_rms']:.2e}, bound used {r['bound_used']:.2f}, "
          f"elements that needed the rms floor {100 * r['floor_share']:.2f} %")
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        json.dump(_LOSS_REC[test], open(os.path.join(d, f"loss_term_errors_{test}.json"), "w"), indent=1)
    except OSError:
        pass
    ceil = LOSS_CEILING.get(test, {}).get(name)
    if ceil is not None:
        assert r["max_err_over_rms"] <= ceil, f"{name}: max err / rms {r['max_err_over_rms']:.3e} > recorded ceiling {ceil:.3e}"
    return r


def _tensor_class(name, shapes):
    """Variable name -> the class of tensor whose error is tracked together (same reduction shape, same kernels): convs by the
    shape of their kernel ([K, Cin, Cout]); a bias joins its conv's class."""
    if name.endswith("/alpha") or name.endswith("/bins"):
        return "quantizer alpha / bins"
    base = name.rsplit("/", 1)[-1]
    if "separable" in name:
        return f"separable conv {base}"
    kshape = shapes.get(name.rsplit("/", 1)[0] + "/kernel")
    K, Ci, Co = kshape
    role = {(1, 20): "block 1x1", (15, 20): "block k15 gate", (9, 20): "block k9"}.get((K, Co if K != 9 else Ci))
    if role is None:
        role = "k55 1 -> C" if (K == 55 and Ci == 1) else ("k55 C -> 1" if K == 55 else f"k{K} {Ci} -> {Co} (down-sampling)")
    if K == 1 and Ci == 1:
        role = "block 1x1, one input channel"
    return role + (" bias" if base == "bias" else " kernel")


def _grad_report(mine, g64, g32, what, only=None):
    """Worst max|hip - f64| / max|f64| per tensor class, next to what the same graph delivers in float32 on the CPU.  Printed
    (pytest -s / -rA), written to gpurun_out/grad_ratios_<what>.json, returned as {class: (hip ratio, fp32-CPU ratio, worst name)}.
    The record also carries the RMS error of the class's worst tensor (rms|hip - f64| / rms|f64|): a leaky-relu kink that flips against
    float64 moves a handful of elements by a whole contribution (the max norm sees only that), the rms norm sees the arithmetic."""
    import json, os
    rep = {}
    shapes = {n: tuple(g.shape) for n, g in g64.items()}
    for name, g in g64.items():
        if only is not None and not only(name):
            continue
        a, b = mine[name].reshape(-1).astype(np.float64), g.reshape(-1)
        scale = max(float(np.max(np.abs(b))), 1e-6)
        err = float(np.max(np.abs(a - b))) / scale
        e32 = float(np.max(np.abs(g32[name].reshape(-1) - b))) / scale
        rb = max(float(np.sqrt(np.mean(b * b))), 1e-12)
        rms = float(np.sqrt(np.mean((a - b) ** 2))) / rb
        rms32 = float(np.sqrt(np.mean((g32[name].reshape(-1) - b) ** 2))) / rb
        k = _tensor_class(name, shapes)
        if k not in rep or err > rep[k][0]:
            rep[k] = (err, e32, name, rms, rms32)
    print(f"\nachieved gradient error ({what}; max|hip - f64| / max|f64| per tensor class | the float32 CPU oracle on the same step):")
    for k in sorted(rep):
        print(f"  {k:32s} {rep[k][0]:.3e} | {rep[k][1]:.3e}   worst: {rep[k][2]}")
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        json.dump({k: dict(hip=v[0], fp32_cpu=v[1], worst=v[2], hip_rms=v[3], fp32_cpu_rms=v[4]) for k, v in rep.items()},
                  open(os.path.join(d, f"grad_ratios_{what}.json"), "w"), indent=1)
    except OSError:
        pass
    return rep


# Recorded ceilings (round 4, MI355X; profiles/r04b_grad_ratios_{joint,follower}.json hold the measured values): 4x what was achieved,
# floor 1e-4, so a regression from 5e-6 to 4e-4 fails here even though it would pass the 5e-4 / 4x-fp32-CPU bound above.  Why a floor of
# 1e-4 and not of 2e-5: ONE activation within fp32 noise of the leaky-relu kink takes the other slope as soon as anything upstream rounds
# differently, and that single element moves a bias gradient by its whole contribution.  Measured: the follower step's encoder gradients
# sit at 0.9e-6 of float64 - and at 5.12e-5 (down-sampling conv of scope_2; 1.9e-5 on the k15 kernels) both when the input frames are
# perturbed by 1e-7 relative and when the quantizer forward runs fused in the C -> 1 conv's launch (4-lane instead of 8-lane groups:
# 1e-7 on the soft codes); the forward is equally accurate in all three runs (decoded 4.2e-8 rms of float64).  The float32 CPU oracle is
# at 5.2e-5 on the same tensor.
GRAD_CEILING = {'follower': {'block 1x1 bias': 0.0001,
              'block 1x1 kernel': 0.0001,
              'block 1x1, one input channel bias': 0.0001,
              'block 1x1, one input channel kernel': 0.0001,
              'block k15 gate bias': 0.0001,
              'block k15 gate kernel': 0.0001,
              'block k9 bias': 0.0001,
              'block k9 kernel': 0.0001,
              'k55 1 -> C bias': 0.0001,
              'k55 1 -> C kernel': 0.0001,
              'k55 C -> 1 bias': 0.0001,
              'k55 C -> 1 kernel': 0.0001,
              'k9 100 -> 100 (down-sampling) bias': 0.0001,
              'k9 100 -> 100 (down-sampling) kernel': 0.0001,
              'quantizer alpha / bins': 0.0001,
              'separable conv bias': 0.0001,
              'separable conv depthwise_kernel': 0.0001,
              'separable conv pointwise_kernel': 0.0001},
 'joint': {'block 1x1 bias': 0.0001,
           'block 1x1 kernel': 0.0001,
           'block 1x1, one input channel bias': 0.00049,
           'block 1x1, one input channel kernel': 0.0001,
           'block k15 gate bias': 0.00027,
           'block k15 gate kernel': 0.00026,
           'block k9 bias': 0.00055,
           'block k9 kernel': 0.00053,
           'k55 1 -> C bias': 0.0001,
           'k55 1 -> C kernel': 0.00026,
           'k55 C -> 1 bias': 0.0001,
           'k55 C -> 1 kernel': 0.0001,
           'k9 100 -> 100 (down-sampling) bias': 0.0001,
           'k9 100 -> 100 (down-sampling) kernel': 0.0001,
           'quantizer alpha / bins': 0.0001,
           'separable conv bias': 0.0001,
           'separable conv depthwise_kernel': 0.0001,
           'separable conv pointwise_kernel': 0.0001}}


def _check_ceilings(rep, what):
    over = [f"{k}: {v[0]:.3e} > recorded ceiling {GRAD_CEILING[what][k]:.3e}" for k, v in rep.items()
            if k in GRAD_CEILING[what] and v[0] > GRAD_CEILING[what][k]]
    assert not over, f"gradient error regressed ({what} step):\n" + "\n".join(over)


# ---- the numerics gate of the split-operand arithmetic (fixed criteria, in the suite) ----
# Per gradient class the rms error of the DEFAULT engine (bf16 matrix cores on operands split into three bf16 pieces, six products) against
# float64 must stay within 1.5x of the reference error + 3e-6, where the reference error is what the EXACT-fp32-instruction engine
# delivers on the same step - the larger of two runs: the same inputs, and the inputs moved by one part in 1e7.  Why the second run: a
# gradient of this network is not a smooth function of its inputs at fp32 resolution - one leaky-relu pre-activation within rounding of
# zero takes the other slope than float64 and moves a whole bias gradient by its contribution, in quanta of ~5e-5 (GRAD_CEILING comment
# above) - so the exact arm's error on ONE input is a sample, not a bound, and a 1e-7 perturbation (far below what any arithmetic could
# resolve) shows the spread.  That is the whole criterion: no clause refers to the float32 CPU oracle.
GATE_FACTOR, GATE_FLOOR, GATE_PERTURB = 1.5, 3e-6, 1e-7


def _class_rms(mine, g64, only=None):
    shapes = {k: v.shape for k, v in g64.items()}
    rep = {}
    for name, g in g64.items():
        if only is not None and not only(name):
            continue
        b = g.reshape(-1)
        rb = float(np.sqrt(np.mean(b ** 2)))
        if rb == 0.0:
            continue
        k = _tensor_class(name, shapes)
        rep[k] = max(rep.get(k, 0.0), float(np.sqrt(np.mean((mine[name].reshape(-1) - b) ** 2))) / rb)
    return rep


def _numerics_gate(what, split_rep, run_exact, g64, only=None):
    """run_exact(perturb) -> named gradients of the exact-arithmetic engine on the (perturbed) inputs."""
    ref = {}
    for pert in (0.0, GATE_PERTURB):
        for k, v in _class_rms(run_exact(pert), g64, only).items():
            ref[k] = max(ref.get(k, 0.0), v)
    lines, over = [], []
    for k in sorted(split_rep):
        lim = GATE_FACTOR * ref[k] + GATE_FLOOR
        lines.append(f"  {k:36s} split {split_rep[k]:.3e}   exact (max of two runs) {ref[k]:.3e}   ratio {split_rep[k] / max(ref[k], 1e-30):5.2f}")
        if split_rep[k] > lim:
            over.append(f"{k}: rms error {split_rep[k]:.3e} > {GATE_FACTOR} x {ref[k]:.3e} + {GATE_FLOOR}")
    print(f"\nnumerics gate ({what} step, B = {B}; rms|g - f64| / rms|f64| of the worst tensor per class):\n" + "\n".join(lines))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, f"numerics_gate_{what}.txt"), "w").write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not over, f"numerics gate of the split arithmetic ({what} step):\n" + "\n".join(over)


def _exact_engine():
    """A second engine on the exact fp32 matrix instruction everywhere (what `bench.py --arith exact` times)."""
    _, eng, _, _ = _setup()
    eng.split_fwd = eng.split_wgrad_arith = eng.split_conv = False
    return eng


def _perturbed(x, pert, seed=99):
    if pert == 0.0:
        return x
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (x * (1.0 + pert * torch.randn(x.shape, generator=g).to(x.device))).contiguous()


def _oracle_joint_step(x_np, lpc_np, ps, dtype):
    """The config-3 joint step on the PyTorch-CPU oracle: decoded frames, the four loss terms per frame, every gradient."""
    import bench
    from oracle import nsc_oracle_torch as OT
    tp = OT.TorchParams(ps, dtype=dtype)
    x = torch.tensor(np.ascontiguousarray(x_np.transpose(0, 2, 1)), dtype=dtype)
    lpc = torch.tensor(lpc_np, dtype=dtype)
    outs, dec = OT.cascade_forward(x, tp, bench.BKD, [[2], [2]], 1.0, True, bench.RES_SCALAR, True)
    pl, _ = OT.scalar_softmax_quantization(lpc, tp.t["lpc_quan/alpha"], tp.t["lpc_quan/bins"], 1.0, True)
    tgt = x[:, :, 0]
    OT.total_loss_sum(dec, tgt, [o["p"] for o in outs], bench.COEFF, 0.0, "finetune_lpc", (pl,)).backward()
    terms = dict(time=OT.mse_loss(dec, tgt), freq=OT.mfcc_loss(dec, tgt), quan=[OT.quan_loss(o["p"]) for o in outs],
                 quan_lpc=OT.quan_loss(pl))
    grads = {k: (t.grad.numpy().astype(np.float64) if t.grad is not None else np.zeros(tuple(t.shape))) for k, t in tp.t.items()}
    return dec.detach().numpy(), terms, grads


def test_headline_step_values_match_the_float64_oracle():
    """The bench workload itself (config 3: 2 codecs on the fed residual + LSF quantizer, joint finetune_lpc step, B = 128,
    bench.synth_batch inputs) against the float64 PyTorch-CPU oracle: decoded frames elementwise, the four loss terms per
    frame, and every gradient tensor (bound: 5e-4 of the tensor's max, or 4x what the same graph delivers in float32 on the
    CPU).  A softer alpha than the reference's -300 keeps the quantizer gradients non-degenerate."""
    import bench
    from tests._util import assert_close, make_store
    _, eng, x, lpc = _setup()
    _, _, x_np, lpc_np = bench.synth_batch(B, 0, torch.device("cuda", 0))
    ps = make_store(2, [[2], [2]], [32, 32], rand_bias=True, alpha=-20.0, lpc=True)
    ps.params["lpc_quan/alpha"] = np.array(-40.0)
    eng.load_named(ps.params)
    eng.refresh_wt()
    eng.grads.zero_()
    dec = eng.forward(x, 1.0, True, lpc_x=lpc)
    c = bench.COEFF
    terms = eng.loss_backward(x, c[0], c[1], [c[2], c[2]], [0.0, 0.0], [True, True], c_quan_lpc=c[2], train_lpc=True)
    torch.cuda.synchronize()
    d64, t64, g64 = _oracle_joint_step(x_np, lpc_np, ps, torch.float64)
    _, _, g32 = _oracle_joint_step(x_np, lpc_np, ps, torch.float32)
    _term("joint", "decoded", dec.cpu().numpy()[:, 0], d64)
    _term("joint", "time loss per frame", terms["time"].cpu().numpy(), t64["time"].detach().numpy())
    _term("joint", "mel loss per frame", terms["freq"].cpu().numpy(), t64["freq"].detach().numpy())
    for i in range(2):
        _term("joint", f"quan loss of codec {i + 1} per frame", terms["quan"][i].cpu().numpy(), t64["quan"][i].detach().numpy())
    _term("joint", "LSF quan loss per frame", terms["quan_lpc"].cpu().numpy(), t64["quan_lpc"].detach().numpy())
    mine = eng.named("grads")
    fails = []
    for name, g in g64.items():
        a, b = mine[name].reshape(-1), g.reshape(-1)
        scale = max(float(np.max(np.abs(b))), 1e-6)
        err = float(np.max(np.abs(a - b))) / scale
        lim = max(5e-4, 4.0 * float(np.max(np.abs(g32[name].reshape(-1) - b))) / scale)
        if not np.all(np.isfinite(a)) or err > lim:
            fails.append(f"{name}: rel err {err:.3e} > {lim:.3e}")
    assert not fails, "gradient mismatches at B = 128:\n" + "\n".join(fails)
    _check_ceilings(_grad_report(mine, g64, g32, "joint"), "joint")
    if eng.split_fwd or eng.split_wgrad_arith or eng.split_conv:
        ex = _exact_engine()
        ex.load_named(ps.params)

        def run_exact(pert):
            xp = _perturbed(x, pert)
            ex.refresh_wt()
            ex.grads.zero_()
            ex.forward(xp, 1.0, True, lpc_x=lpc)
            ex.loss_backward(xp, c[0], c[1], [c[2], c[2]], [0.0, 0.0], [True, True], c_quan_lpc=c[2], train_lpc=True)
            torch.cuda.synchronize()
            return ex.named("grads")
        _numerics_gate("joint", _class_rms(mine, g64), run_exact, g64)


def test_headline_forward_at_alpha_minus_300_matches_the_float64_oracle():
    """The quantizer regime the bench and the reference START in (constants.py:5: alpha = -300, bins linspace(-1, 1, 32)), at
    the full B = 128 of config 3: decoded frames, the codes, the soft assignment p and the four loss terms of the training
    forward (the_share = 1) against the float64 oracle; then the hard forward (the_share = 0): codes BIT-EXACT wherever the
    float code is not within fp32 noise of a bin midpoint (a frame with such a code in codec 1 is left out of codec 2's
    comparison: its residual input legitimately differs)."""
    import bench
    from oracle import nsc_oracle_torch as OT
    from tests._util import assert_close, make_store
    _, eng, x, lpc = _setup()
    _, _, x_np, lpc_np = bench.synth_batch(B, 0, torch.device("cuda", 0))
    ps = make_store(2, [[2], [2]], [32, 32], rand_bias=True, alpha=None, lpc=True)        # alpha stays at the reference's -300
    assert float(ps.params["scope_1/alpha"]) == -300.0 and float(ps.params["lpc_quan/alpha"]) == -300.0
    eng.load_named(ps.params)
    eng.refresh_wt()
    tp = OT.TorchParams(ps, dtype=torch.float64)
    xt = torch.tensor(np.ascontiguousarray(x_np.transpose(0, 2, 1)), dtype=torch.float64)
    lt = torch.tensor(lpc_np, dtype=torch.float64)
    tgt = xt[:, :, 0]
    with torch.no_grad():
        # ---- training forward: soft assignment ----
        dec = eng.forward(x, 1.0, True, lpc_x=lpc, want_p=True)
        c = bench.COEFF
        eng.grads.zero_()
        terms = eng.loss_backward(x, c[0], c[1], [c[2], c[2]], [0.0, 0.0], [True, True], c_quan_lpc=c[2], train_lpc=True)
        torch.cuda.synchronize()
        outs, d64 = OT.cascade_forward(xt, tp, bench.BKD, [[2], [2]], 1.0, True, bench.RES_SCALAR, True)
        pl, ql = OT.scalar_softmax_quantization(lt, tp.t["lpc_quan/alpha"], tp.t["lpc_quan/bins"], 1.0, True)
        _term("alpha300", "decoded", dec.cpu().numpy()[:, 0], d64.numpy())
        half_bin = 1.0 / 31.0
        for i, (cd, o) in enumerate(zip(eng.codecs, outs)):
            fc = o["floating_code"].numpy()[:, :, 0]
            assert_close(cd.code.cpu().numpy()[:, 0], fc, what=f"float codes of codec {i + 1}")
            # alpha (b - a) = 19.4 between neighbouring bins: p is a two-bin sigmoid 300 x (distance to the midpoint) wide; an
            # fp32 code error of 1e-6 moves it by 6e-4 of itself at worst - absolute tolerance on p, relative on the codes
            assert float(np.max(np.abs(cd.p.cpu().numpy() - o["p"].numpy()))) <= 2e-3, f"p of codec {i + 1}"
            assert float(np.max(np.abs(cd.qcode.cpu().numpy()[:, 0] - o["code"].numpy()[:, :, 0]))) <= 2e-3 * 2 * half_bin
            _term("alpha300", f"quan loss of codec {i + 1}", terms["quan"][i].cpu().numpy(), OT.quan_loss(o["p"]).numpy(), tol=2e-3)
        _term("alpha300", "time loss per frame", terms["time"].cpu().numpy(), OT.mse_loss(d64, tgt).numpy())
        _term("alpha300", "mel loss per frame", terms["freq"].cpu().numpy(), OT.mfcc_loss(d64, tgt).numpy())
        _term("alpha300", "LSF quan loss per frame", terms["quan_lpc"].cpu().numpy(), OT.quan_loss(pl).numpy(), tol=2e-3)
        # ---- hard forward: nearest-bin codes, bit exact away from the midpoints ----
        dech = eng.forward(x, 1.0, False, lpc_x=lpc)
        torch.cuda.synchronize()
        outh, dh64 = OT.cascade_forward(xt, tp, bench.BKD, [[2], [2]], 1.0, False, bench.RES_SCALAR, True)
        bins64 = tp.t["scope_1/bins"].numpy()
        mids = 0.5 * (bins64[1:] + bins64[:-1])
        ok_frames = np.ones(B, bool)
        n_cmp = 0
        for i, (cd, o) in enumerate(zip(eng.codecs, outh)):
            fc = o["floating_code"].numpy()[:, :, 0]
            safe = np.min(np.abs(fc[:, :, None] - mids[None, None, :]), axis=-1) > 2e-5          # [B, L]
            q_hip = cd.qcode.cpu().numpy()[:, 0]
            q_ref = o["code"].numpy()[:, :, 0].astype(np.float32)
            sel = safe & ok_frames[:, None]
            assert sel.mean() > 0.9, "the midpoint mask must leave almost everything in"
            assert np.array_equal(q_hip[sel], q_ref[sel]), f"hard codes of codec {i + 1} differ away from the bin midpoints"
            n_cmp += int(sel.sum())
            ok_frames &= safe.all(axis=1)
        assert ok_frames.sum() >= B // 2
        assert_close(dech.cpu().numpy()[ok_frames, 0], dh64.numpy()[ok_frames], what="decoded, hard codes, frames away from midpoints")
        print(f"\nalpha -300, B = 128: {n_cmp} hard codes compared bit-exactly, {int(ok_frames.sum())} of {B} frames in the decoded comparison")


def test_follower_step_values_match_the_float64_oracle():
    """The second step of config 3 (bench.py --follower; cmrl.py:137-293): codec 1 frozen - it runs forward-only and keeps
    no activations -, codec 2 trains on its residual with its own quan + entropy terms (tau 0.3, histogram of the batch).
    Decoded frames, loss terms and every gradient of scope_2 against the float64 oracle at B = 128; scope_1 and the LSF
    quantizer get exactly zero."""
    import bench
    from oracle import nsc_oracle_torch as OT
    from tests._util import assert_close, make_store
    _, eng, x, lpc = _setup()
    _, _, x_np, lpc_np = bench.synth_batch(B, 0, torch.device("cuda", 0))
    ps = make_store(2, [[2], [2]], [32, 32], rand_bias=True, alpha=-20.0, lpc=True)
    eng.load_named(ps.params)
    c, tau = bench.COEFF, 0.3
    cfg = dict(is_quan_on=1.0, c_time=c[0], c_freq=c[1], c_quan=[0.0, c[2]], c_ent=[0.0, tau], trainable=[False, True], lr=0.0, slot=1,
               quan_op=True)
    eng.refresh_wt()
    eng.grads.zero_()
    dec = eng.forward(x, 1.0, True, lpc_x=lpc, first_needed=1)
    terms = eng.loss_backward(x, c[0], c[1], cfg["c_quan"], cfg["c_ent"], cfg["trainable"])
    torch.cuda.synchronize()

    def oracle(dtype):
        tp = OT.TorchParams(ps, dtype=dtype)
        xt = torch.tensor(np.ascontiguousarray(x_np.transpose(0, 2, 1)), dtype=dtype)
        outs, d = OT.cascade_forward(xt, tp, bench.BKD, [[2], [2]], 1.0, True, bench.RES_SCALAR, True)
        tgt = xt[:, :, 0]
        OT.total_loss_sum(d, tgt, [o["p"] for o in outs], c, tau, "quan_last").backward()
        g = {k: (t.grad.numpy().astype(np.float64) if t.grad is not None else np.zeros(tuple(t.shape))) for k, t in tp.t.items()}
        return d.detach().numpy(), outs, g
    d64, outs64, g64 = oracle(torch.float64)
    _, _, g32 = oracle(torch.float32)
    assert_close(dec.cpu().numpy()[:, 0], d64, what="decoded, follower step")
    assert_close(terms["quan"][1].cpu().numpy(), OT.quan_loss(outs64[1]["p"]).detach().numpy(), what="quan loss of codec 2 per frame")
    mine = eng.named("grads")
    fails = []
    for name, g in g64.items():
        a, b = mine[name].reshape(-1), g.reshape(-1)
        if not name.startswith("scope_2/"):
            if np.any(a != 0.0):
                fails.append(f"{name}: frozen, but its gradient is not zero")
            continue
        scale = max(float(np.max(np.abs(b))), 1e-6)
        err = float(np.max(np.abs(a - b))) / scale
        lim = max(5e-4, 4.0 * float(np.max(np.abs(g32[name].reshape(-1) - b))) / scale)
        if not np.all(np.isfinite(a)) or err > lim:
            fails.append(f"{name}: rel err {err:.3e} > {lim:.3e}")
    assert not fails, "follower-step gradient mismatches at B = 128:\n" + "\n".join(fails)
    _check_ceilings(_grad_report(mine, g64, g32, "follower", only=lambda n: n.startswith("scope_2/")), "follower")
    if eng.split_fwd or eng.split_wgrad_arith or eng.split_conv:
        ex = _exact_engine()
        ex.load_named(ps.params)
        only2 = lambda n: n.startswith("scope_2/")

        def run_exact(pert):
            xp = _perturbed(x, pert)
            ex.refresh_wt()
            ex.grads.zero_()
            ex.forward(xp, 1.0, True, lpc_x=lpc, first_needed=1)
            ex.loss_backward(xp, c[0], c[1], cfg["c_quan"], cfg["c_ent"], cfg["trainable"])
            torch.cuda.synchronize()
            return ex.named("grads")
        _numerics_gate("follower", _class_rms(mine, g64, only2), run_exact, g64, only2)


def test_config4_forward_values_match_the_float64_oracle():
    """BASELINE config 4 at its full per-GPU batch: 4 codecs, each with two down-/up-sampling stages ('2 2': 128 codes, blocks
    at C = 100, 50 and 25), B = 256, forward with hard codes - decoded frames and every codec's codes vs the float64 oracle
    run on a 32-frame slice (frames are independent: the engine's result for those frames must not depend on the other 224)."""
    import bench
    from nsc_amd.engine import CascadeEngine
    from oracle import nsc_oracle_torch as OT
    from tests._util import assert_close, make_store
    dev = torch.device("cuda", 0)
    Bc, N, sl = 256, 4, 32
    st, nb = [[2, 2]] * N, [32] * N
    ps = make_store(N, st, nb, rand_bias=True, alpha=-20.0)
    eng = CascadeEngine(Bc, N, bench.BKD, st, nb, res_scalar=2.0, device=dev)
    eng.load_named(ps.params)
    x, _, x_np, _ = bench.synth_batch(Bc, 0, dev)
    dec = eng.forward(x, 1.0, True, want_p=True)
    torch.cuda.synchronize()
    tp = OT.TorchParams(ps)
    xt = torch.tensor(np.ascontiguousarray(x_np[:sl].transpose(0, 2, 1)), dtype=torch.float64)
    with torch.no_grad():
        outs, d64 = OT.cascade_forward(xt, tp, bench.BKD, st, 1.0, True, 2.0, False)
    assert eng.codecs[-1].L == 128
    assert_close(dec.cpu().numpy()[:sl, 0], d64.numpy(), what="config 4 decoded (soft assignment)")
    for i, c in enumerate(eng.codecs):
        assert_close(c.code.cpu().numpy()[:sl, 0], outs[i]["floating_code"].numpy()[:, :, 0], what=f"config 4 code of codec {i + 1}",
                     atol=1e-6)
        assert_close(c.p.cpu().numpy()[:sl], outs[i]["p"].numpy(), what=f"config 4 soft assignment of codec {i + 1}", atol=1e-6)
    # hard codes: a code within fp32 noise of a bin midpoint may legitimately pick the other bin than the float64 oracle, so
    # the value check above is the soft one; the hard forward is bit-compared between batch sizes in tests/test_configs_gpu.py
    assert bool(torch.isfinite(eng.forward(x, 1.0, False)).all())
