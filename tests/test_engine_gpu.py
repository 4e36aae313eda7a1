"""GPU parity of the whole codec / cascade engine (forward, losses, gradients, Adam) against the CPU oracle
and the committed golden fixture.  fp32 HIP vs float64 oracle: 1e-4 rel (north_star), gradients 5e-4 of the
largest gradient entry of each tensor group (they pass through ~40 fp32 layers)."""
import os

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT
from tests._util import BKD, assert_close, dev, make_store, relerr, synth_frames

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _engine(B, N, strides, bins, ps, **kw):
    from nsc_amd.engine import CascadeEngine
    eng = CascadeEngine(B, N, BKD, strides, bins, **kw)
    eng.load_named(ps.params)
    eng.refresh_wt()
    return eng


def _oracle_grads(ps, x, N, strides, coeff, tau, mode, is_quan_on, rs=1.0, scale_first=False, lpc_x=None, trainable=None):
    tp = OT.TorchParams(ps)
    xt = torch.tensor(x)
    outs, dec = OT.cascade_forward(xt, tp, BKD, strides, is_quan_on, True, rs, scale_first)
    p_extra = ()
    if lpc_x is not None:
        pl, _ = OT.scalar_softmax_quantization(torch.tensor(lpc_x), tp.t["lpc_quan/alpha"], tp.t["lpc_quan/bins"], is_quan_on, True)
        p_extra = (pl,)
    loss = OT.total_loss_sum(dec, xt[:, :, 0], [o["p"] for o in outs], coeff, tau, mode, p_extra)
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else np.zeros(t.shape)) for k, t in tp.t.items()}
    # the same graph in float32 on the CPU calibrates what fp32 arithmetic can deliver for each tensor
    tp32 = OT.TorchParams(ps, dtype=torch.float32)
    x32 = torch.tensor(x, dtype=torch.float32)
    o32, d32 = OT.cascade_forward(x32, tp32, BKD, strides, is_quan_on, True, rs, scale_first)
    pe32 = ()
    if lpc_x is not None:
        pe32 = (OT.scalar_softmax_quantization(torch.tensor(lpc_x, dtype=torch.float32), tp32.t["lpc_quan/alpha"],
                                               tp32.t["lpc_quan/bins"], is_quan_on, True)[0],)
    OT.total_loss_sum(d32, x32[:, :, 0], [o["p"] for o in o32], coeff, tau, mode, pe32).backward()
    for k, t in tp32.t.items():
        g32 = t.grad.numpy().astype(np.float64) if t.grad is not None else np.zeros(t.shape)
        scale = max(np.max(np.abs(grads[k])), 1e-6)
        FP32_ERR[k] = float(np.max(np.abs(g32 - grads[k])) / scale)
    return outs, dec.detach().numpy(), float(loss.detach()), grads


FP32_ERR = {}


def _check_grads(eng, grads, scopes, tol=5e-4):
    """Per tensor: max |hip - f64| / max |f64| <= max(5e-4, 4 x the error of the float32 CPU oracle)."""
    mine = eng.named("grads")
    fails = []
    for name, g in grads.items():
        if not any(name.startswith(s + "/") for s in scopes):
            continue
        a, b = mine[name].reshape(-1), np.asarray(g).reshape(-1)
        scale = max(np.max(np.abs(b)), 1e-6)
        err = np.max(np.abs(a - b)) / scale
        assert np.all(np.isfinite(a)), name
        lim = max(tol, 4.0 * FP32_ERR.get(name, 0.0))
        if err > lim:
            fails.append(f"{name}: rel err {err:.3e} > {lim:.3e} (scale {scale:.3e}, cpu-fp32 err {FP32_ERR.get(name, 0):.3e})")
    assert not fails, "gradient mismatches:\n" + "\n".join(fails)


def test_single_codec_forward_matches_golden_and_oracle():
    gold = np.load(os.path.join(GOLD, "codec_golden.npz"))
    from tests.test_oracle import _setup
    ps, x = _setup([2], B=2)
    assert np.array_equal(x, gold["x"])
    eng = _engine(2, 1, [[2]], [32], ps)
    dec = eng.forward(dev(x.transpose(0, 2, 1)), 1.0, True, want_p=True)
    torch.cuda.synchronize()
    c = eng.codecs[0]
    assert_close(c.code.cpu().numpy()[:, 0, :, None], gold["floating_code"], what="floating code vs golden")
    assert_close(dec.cpu().numpy()[:, 0], gold["decoded"], what="decoded vs golden")
    assert_close(c.hist.cpu().numpy(), gold["p_sum_hist"], what="histogram vs golden")
    assert_close(c.quan.cpu().numpy(), gold["quan_loss"], what="quan loss vs golden")
    terms = eng.loss_backward(dev(x.transpose(0, 2, 1)), 60.0, 10.0, [10.0], [0.3], [True])
    torch.cuda.synchronize()
    assert_close(terms["time"].cpu().numpy(), gold["time_loss"], what="time loss vs golden")
    assert_close(terms["freq"].cpu().numpy(), gold["freq_loss"], what="freq loss vs golden")     # (the standard 1e-4 bound: round 6)
    assert abs(float(terms["ent"][0]) - float(gold["ent_loss"])) < 1e-4 * float(gold["ent_loss"])
    flat = eng.grads.cpu().numpy()
    assert_close(flat, gold["flat_grad"], tol=5e-4, what="flat gradient vs golden")


@pytest.mark.parametrize("strides", [[2], [2, 2]])
def test_single_codec_quan_step_gradients(strides):
    B = 3
    ps = make_store(1, [strides], [32])
    x = synth_frames(B)
    coeff, tau = [60.0, 10.0, 10.0, 0.0], 0.4
    outs, dec, loss, grads = _oracle_grads(ps, x, 1, [strides], coeff, tau, "quan_last", 1.0)
    eng = _engine(B, 1, [strides], [32], ps)
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    d = eng.forward(xd, 1.0, True)
    terms = eng.loss_backward(xd, coeff[0], coeff[1], [coeff[2]], [tau], [True])
    torch.cuda.synchronize()
    assert_close(d.cpu().numpy()[:, 0], dec, what="decoded")
    assert_close(eng.codecs[0].code.cpu().numpy()[:, 0], outs[0]["floating_code"].detach().numpy()[:, :, 0], what="code")
    _check_grads(eng, grads, ["scope_1"])


def test_pretrain_no_quan_step():
    """config 1 plumbing: is_quan_on=0, loss_no_quan; alpha/bins must receive exactly zero gradient."""
    B = 2
    ps = make_store(1, [[2]], [32], alpha=None)
    x = synth_frames(B)
    coeff = [60.0, 10.0, 10.0, 0.0]
    outs, dec, loss, grads = _oracle_grads(ps, x, 1, [[2]], coeff, 0.0, "no_quan", 0.0)
    eng = _engine(B, 1, [[2]], [32], ps)
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    d = eng.forward(xd, 0.0, True)
    eng.loss_backward(xd, coeff[0], coeff[1], [0.0], [0.0], [True])
    torch.cuda.synchronize()
    assert_close(d.cpu().numpy()[:, 0], dec, what="decoded")
    _check_grads(eng, grads, ["scope_1"])
    g = eng.named("grads")
    assert float(np.abs(g["scope_1/alpha"]).max()) == 0.0 and float(np.abs(g["scope_1/bins"]).max()) == 0.0


@pytest.mark.parametrize("switch", ["fused_up", "batch_conv_wgrad", "batch_wgrad", "fused_dgrad", "fused_fwd", "fused_chain", "fused_pairs", "fused_quant"])
def test_composed_paths_equal_the_fused_ones(switch):
    """Every fused / batched launch has a composed per-op path behind an engine switch (other widths and dilations take it):
    with the switch off, one joint 2-codec step ('2 2' codecs: C = 100 and 50 up-sampling stages) gives the same decoded
    frames and the same gradients up to float32 summation order."""
    B = 2
    ps = make_store(2, [[2, 2], [2, 2]], [32, 32])
    x = dev(synth_frames(B).transpose(0, 2, 1))
    res = []
    for on in (True, False):
        eng = _engine(B, 2, [[2, 2], [2, 2]], [32, 32], ps)
        setattr(eng, switch, on)
        eng.grads.zero_()
        d = eng.forward(x, 1.0, True).clone()
        eng.loss_backward(x, 60.0, 10.0, [10.0, 10.0], [0.3, 0.5], [True, True])
        torch.cuda.synchronize()
        res.append((d.cpu().numpy(), eng.named("grads")))
    assert_close(res[1][0], res[0][0], tol=2e-5, what=f"decoded, {switch} off vs on")
    for name, g in res[0][1].items():
        a, b = res[1][1][name].reshape(-1), g.reshape(-1)
        scale = max(float(np.max(np.abs(b))), 1e-6)
        assert float(np.max(np.abs(a - b))) <= 2e-4 * scale, (switch, name, float(np.max(np.abs(a - b))) / scale)


@pytest.mark.parametrize("rs", [1.0, 2.0])
def test_two_codec_finetune_and_follower(rs):
    B = 2
    ps = make_store(2, [[2], [2]], [32, 32])
    x = synth_frames(B)
    coeff, tau = [60.0, 10.0, 10.0, 0.0], [0.3, 0.5]
    outs, dec, loss, grads = _oracle_grads(ps, x, 2, [[2], [2]], coeff, tau, "finetune", 1.0, rs=rs)
    eng = _engine(B, 2, [[2], [2]], [32, 32], ps, res_scalar=rs)
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    d = eng.forward(xd, 1.0, True)
    # cmrl.py:355: the finetune quan term is a batch-summed scalar broadcast into the [B] loss => weight coeff[2] * B
    eng.loss_backward(xd, coeff[0], coeff[1], [coeff[2] * B] * 2, tau, [True, True])
    torch.cuda.synchronize()
    assert_close(d.cpu().numpy()[:, 0], dec, what="cascade decoded")
    _check_grads(eng, grads, ["scope_1", "scope_2"])
    # follower: codec 1 frozen, newest codec only (cmrl.py:95-113)
    outs, dec, loss, grads = _oracle_grads(ps, x, 2, [[2], [2]], coeff, [0.5], "quan_last", 1.0, rs=rs)
    eng.grads.zero_()
    eng.forward(xd, 1.0, True)
    eng.loss_backward(xd, coeff[0], coeff[1], [0.0, coeff[2]], [0.0, 0.5], [False, True])
    torch.cuda.synchronize()
    _check_grads(eng, grads, ["scope_2"])
    a, b = eng.layout.scope_range("scope_1")
    assert float(eng.grads[a:b].abs().max()) == 0.0


def test_two_codec_lpc_finetune():
    """config 3 (north-star step): LPC residual fed as the input, LSF quantizer (16 x 256 bins) only through
    quan_loss, every codec scaled by res_scalar, no entropy term (cmrl.py:392-511)."""
    B = 2
    ps = make_store(2, [[2], [2]], [32, 32], lpc=True)
    # seed 1234 puts one pre-activation of codec 2's up-sampler at |z| ~ 1e-8, i.e. ON the leaky-relu kink: the
    # float64 oracle and any float32 run may then legitimately pick different slopes for that element (measured:
    # a one-off probe in round 2).  Use a seed without such an element; kink handling itself is unit-tested per kernel.
    x = synth_frames(B, seed=4321)
    rng = np.random.default_rng(3)
    lpc_x = np.sort(rng.uniform(0.03, 3.1, (B, 16, 1)), axis=1).astype(np.float32).astype(np.float64)
    coeff = [60.0, 10.0, 10.0, 0.0]
    ps.params["lpc_quan/alpha"] = np.array(-40.0)
    outs, dec, loss, grads = _oracle_grads(ps, x, 2, [[2], [2]], coeff, 0.0, "finetune_lpc", 1.0, rs=2.0,
                                           scale_first=True, lpc_x=lpc_x)
    eng = _engine(B, 2, [[2], [2]], [32, 32], ps, res_scalar=2.0, scale_first=True, lpc=True)
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    d = eng.forward(xd, 1.0, True, lpc_x=dev(lpc_x))
    eng.loss_backward(xd, coeff[0], coeff[1], [coeff[2]] * 2, [0.0, 0.0], [True, True], c_quan_lpc=coeff[2])
    torch.cuda.synchronize()
    assert_close(d.cpu().numpy()[:, 0], dec, what="lpc cascade decoded")
    _check_grads(eng, grads, ["scope_1", "scope_2", "lpc_quan"])


def test_three_adam_steps_track_the_oracle():
    B = 2
    ps = make_store(1, [[2]], [32])
    x = synth_frames(B)
    coeff, tau, lr = [60.0, 10.0, 10.0, 0.0], 0.2, 2e-4
    eng = _engine(B, 1, [[2]], [32], ps)
    xd = dev(x.transpose(0, 2, 1))
    tp = OT.TorchParams(ps)
    plist = [tp.t[k] for k in tp.names]
    ms = [torch.zeros_like(p) for p in plist]
    vs = [torch.zeros_like(p) for p in plist]
    for t in range(1, 4):
        for p in plist:
            p.grad = None
        outs, dec = OT.cascade_forward(torch.tensor(x), tp, BKD, [[2]], 1.0, True)
        OT.total_loss_sum(dec, torch.tensor(x)[:, :, 0], [outs[0]["p"]], coeff, tau, "quan_last").backward()
        OT.adam_tf1_step_(plist, [p.grad for p in plist], ms, vs, t, lr)
        eng.grads.zero_()
        eng.refresh_wt()
        eng.forward(xd, 1.0, True)
        eng.loss_backward(xd, coeff[0], coeff[1], [coeff[2]], [tau], [True])
        eng.adam_step(["scope_1"], lr, slot=1)
    torch.cuda.synchronize()
    mine = eng.named("params")
    # Adam's first steps move every weight by ~lr whatever the gradient scale, so an entry whose gradient is
    # below fp32 noise can legitimately flip sign: compare distributions, not the max norm (the kernel itself
    # is checked exactly in test_kernels_gpu.test_adam_tf1).
    bad = tot = 0
    for k in tp.names:
        ref = tp.t[k].detach().numpy()
        d = np.abs(mine[k] - ref)
        bad += int((d > 0.1 * lr).sum())
        tot += d.size
        assert np.median(d) < 0.01 * lr, (k, float(np.median(d)))
    assert bad / tot < 0.02, (bad, tot)


_DDP_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
from nsc_amd.dist import Comm
from nsc_amd.engine import CascadeEngine
from tests._util import BKD, make_store, synth_frames, dev
comm = Comm(backend=%(backend)r)                # gloo: two ranks share the one GPU of the test box; nccl: one GPU per rank
B, Bl = 4, 2
LPC = %(lpc)r
ps = make_store(2, [[2], [2]], [32, 32], lpc=LPC)
x = synth_frames(B)
cfg = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0, 10.0], c_ent=[0.3, 0.5], trainable=[True, True], lr=2e-4, slot=1)
kw = {}
lpc_all = None
if LPC:     # the north-star step: fed LPC residual, every codec scaled, LSF quantizer trained through its quan term
    ps.params["lpc_quan/alpha"] = np.array(-40.0)
    cfg.update(c_quan_lpc=10.0, train_lpc=True, quan_op=True)
    kw = dict(res_scalar=2.0, scale_first=True, lpc=True)
    lpc_all = np.sort(np.random.default_rng(3).uniform(0.03, 3.1, (B, 16, 1)), axis=1).astype(np.float32)
lo, hi = comm.shard(B)
eng = CascadeEngine(Bl, 2, BKD, [[2], [2]], [32, 32], **kw)
eng.load_named(ps.params)
xd = dev(x[lo:hi].transpose(0, 2, 1))
lx = dev(lpc_all[lo:hi]) if LPC else None
# Two ranks SHARE the one GPU of the test box here (gloo).  Until round 6 that made 1-2 %% of the steps differ in a few entries (packed
# fp32 vector instructions under a shared GPU: LAB_NOTES.md, profiles/r06_gpu_sharing_*.txt; the library is built without them now and
# test_steps_are_bit_stable_when_two_processes_share_the_gpu guards it).  Every quantity below is still the elementwise MEDIAN of three
# repetitions - what the test got before the cause was known: a transient of one repetition drops out, a systematic difference stays.
def med3(run):
    outs = [run() for _ in range(3)]
    return tuple(np.median(np.stack([o[k] for o in outs]), axis=0) for k in range(len(outs[0])))


def eager_step():
    eng.load_named(ps.params); eng.reset_adam()
    eng.train_step(xd, xd, cfg, lpc_x=lx, comm=comm)   # gradient all-reduce(s) + histograms (eager launches)
    torch.cuda.synchronize()
    return eng.grads.cpu().numpy(), eng.params.cpu().numpy(), np.array([float(c.ent.item()) for c in eng.codecs])


g_eager, p_eager, ent_eager = med3(eager_step)          # the default layout: one message at the tail of the step
ent_eager = [float(v) for v in ent_eager]
# the same step replayed as hipGraph SEGMENTS cut at the collectives (what bench.py does at N > 1), in both message layouts
seg = {}
for overlap in (True, False):
    eng.dp_overlap = overlap
    g_lay = med3(eager_step)[0]                                # eager step of THIS layout
    eng.load_named(ps.params); eng.reset_adam()
    st = eng.capture_train_step(xd, xd, cfg, lpc_x=lx, comm=comm)

    def replay_step():
        eng.load_named(ps.params); eng.reset_adam()
        st.replay()
        torch.cuda.synchronize()
        return eng.grads.cpu().numpy(), eng.params.cpu().numpy()
    g_rep, p_rep = med3(replay_step)
    seg[overlap] = (g_rep, p_rep, st.nseg, st.ncoll, float(np.abs(g_rep - g_lay).max() / np.abs(g_lay).max()))
    st.replay(); st.replay()                                  # replays keep working (Adam's device step counter advances)
    torch.cuda.synchronize()
if comm.rank == 0:
    ref = CascadeEngine(B, 2, BKD, [[2], [2]], [32, 32], **kw)
    ref.load_named(ps.params)
    xf = dev(x.transpose(0, 2, 1))
    ref.train_step(xf, xf, cfg, lpc_x=dev(lpc_all) if LPC else None)
    torch.cuda.synchronize()
    g2, p2 = ref.grads.cpu().numpy(), ref.params.cpu().numpy()
    ge = lambda g: float(np.abs(g - g2).max() / np.abs(g2).max())
    pe = lambda p_: float(np.mean(np.abs(p_ - p2) > 1e-6))
    print(json.dumps({"gerr": ge(g_eager), "perr": pe(p_eager), "ent": ent_eager,
                      "ent_ref": [float(e.item()) for e in (c.ent for c in ref.codecs)],
                      # [4]: the replay against the EAGER step of the same message layout (same kernels, same messages)
                      "seg_overlap": [ge(seg[True][0]), pe(seg[True][1]), seg[True][2], seg[True][3], seg[True][4]],
                      "seg_tail": [ge(seg[False][0]), pe(seg[False][1]), seg[False][2], seg[False][3], seg[False][4]]}))
comm.barrier()
comm.close()
'''


def test_data_parallel_engine_two_ranks_over_rccl(tmp_path):
    """The same two-rank run over RCCL (backend nccl, one GPU per rank): skipped on a one-GPU box (the builder's test boxes
    have one; the driver's multi-GPU node is where this runs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    test_data_parallel_engine_two_ranks_equals_one_process(tmp_path, True, backend="nccl", port="29637")


@pytest.mark.parametrize("lpc", [False, True])
def test_data_parallel_engine_two_ranks_equals_one_process(tmp_path, lpc, backend="gloo", port="29633"):
    """2 ranks x 2 frames (per-scope gradient SUM all-reduces issued during the backward pass + global-batch entropy histogram)
    == 1 process x 4 frames, on the HIP path; lpc: the config-3 step, whose LSF-quantizer gradients ride in scope_1's message."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER % {"root": root, "lpc": bool(lpc), "backend": backend})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", port, str(script)], capture_output=True, text=True, env=env,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["gerr"] < 2e-4, res                      # same gradient up to fp32 reduction order
    # and the same Adam step: step 1 moves every weight by ~lr*sign(g), so only weights whose gradient is at the fp32
    # reduction-order noise floor may differ (a vanishing fraction)
    assert res["perr"] < 2e-3, res
    assert np.allclose(res["ent"], res["ent_ref"], rtol=1e-5), res   # entropy is that of the GLOBAL batch on every rank
    # segmented hipGraph replay == the eager data-parallel step (same kernels, same messages; a few gradients are float
    # atomics, so not bit for bit) == one process
    assert res["seg_overlap"][0] < 2e-4 and res["seg_overlap"][1] < 2e-3 and res["seg_overlap"][4] < 1e-5, res
    assert res["seg_tail"][0] < 2e-4 and res["seg_tail"][1] < 2e-3 and res["seg_tail"][4] < 1e-5, res
    # forward + loss | histograms | codec 2 | codec 1 (+ LSF quantizer) | wait | Adam   vs  one gradient message at the tail
    assert res["seg_overlap"][3] == 3 and res["seg_tail"][3] == 2, res        # histograms + two scopes | histograms + one
    assert res["seg_overlap"][2] == 4 and res["seg_tail"][2] == 3, res        # no empty segment between message and wait


def test_validation_frame_entropies_match_batch_of_one_oracle():
    """engine.frame_entropies (one batched forward + per-frame histograms) == the oracle run one frame at a time with
    the_share=False, is_quan_on=1, as the reference's validation loop does (nsc_module:685-722)."""
    B = 3
    ps = make_store(2, [[2], [2]], [32, 32])
    for i in (1, 2):                                  # a softer alpha so the per-frame histograms are not one-hot
        ps.params[f"scope_{i}/alpha"] = np.array(-8.0)
    x = synth_frames(B)
    eng = _engine(B, 2, [[2], [2]], [32, 32], ps)
    got = eng.frame_entropies(dev(x.transpose(0, 2, 1))).cpu().numpy()
    for b in range(B):
        ps.begin_replay()
        outs, _ = O.cascade_forward(x[b:b + 1].astype(np.float64), ps, BKD, [[2], [2]], [32, 32], 1.0, False)
        ref = [O.entropy_coding_loss(o["p"]) for o in outs]
        assert np.allclose(got[:, b], ref, rtol=2e-4, atol=2e-4), (b, got[:, b], ref)


def test_three_codec_follower_two_downsamplings():
    """BASELINE config 4 topology at test size: codecs with TWO down/up-sampling stages ('2 2': L = 128, blocks at C = 100,
    50 and 25), a third codec trained as follower of two frozen ones (cmrl.py:22-135): decoded sum and the newest codec's
    gradients vs the oracle; frozen scopes get exactly zero gradient."""
    B, N = 2, 3
    st, nb = [[2, 2]] * N, [32] * N
    ps = make_store(N, st, nb)
    x = synth_frames(B)
    coeff = [60.0, 10.0, 10.0, 0.0]
    outs, dec, loss, grads = _oracle_grads(ps, x, N, st, coeff, [0.4], "quan_last", 1.0, rs=1.0)
    eng = _engine(B, N, st, nb, ps)
    xd = dev(x.transpose(0, 2, 1))
    eng.grads.zero_()
    d = eng.forward(xd, 1.0, True)
    eng.loss_backward(xd, coeff[0], coeff[1], [0.0, 0.0, coeff[2]], [0.0, 0.0, 0.4], [False, False, True])
    torch.cuda.synchronize()
    assert eng.codecs[-1].L == 128
    assert_close(d.cpu().numpy()[:, 0], dec, what="3-codec cascade decoded")
    _check_grads(eng, grads, ["scope_3"])
    for s in ("scope_1", "scope_2"):
        a, b = eng.layout.scope_range(s)
        assert float(eng.grads[a:b].abs().max()) == 0.0


def test_parameter_write_after_refresh_invalidates_the_block_images():
    """refresh_wt -> a direct write to the parameters (params.copy_, a view, set_params) -> forward must NOT decode with
    the stale kernel-ready images: the forward has to equal the plain entry points on the NEW weights."""
    B = 4
    ps_a = make_store(1, [[2]], [32], seed=1)
    ps_b = make_store(1, [[2]], [32], seed=2)
    x = synth_frames(B, seed=3)
    xd = dev(x.transpose(0, 2, 1))
    ref = _engine(B, 1, [[2]], [32], ps_b)
    ref.use_images = False
    want = ref.forward(xd, 1.0, True).clone()
    for how in ("copy_", "view", "set_params", "load_named"):
        eng = _engine(B, 1, [[2]], [32], ps_a)             # refresh_wt has run: images valid for weights A
        assert eng.images_valid
        if how == "copy_":
            eng.params.copy_(ref.params)
        elif how == "view":
            for name in eng.layout.entries:
                eng.view(name).copy_(ref.view(name))
        elif how == "set_params":
            eng.set_params(ref.params)
        else:
            eng.load_named(ps_b.params)
        assert not eng.images_valid, how
        got = eng.forward(xd, 1.0, True)
        assert torch.equal(got, want), how
        eng.refresh_wt()
        assert eng.images_valid
        got = eng.forward(xd, 1.0, True)
        if eng.split_fwd:      # the split-operand kernels (NSC_BLOCK_ARITH=split) agree with the plain entry points to fp32 rounding,
            # not bit for bit; stale weights would be off by the size of the signal
            assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), how + " after refresh"
        else:
            assert torch.equal(got, want), how + " after refresh"


def test_pair_launches_are_off_under_per_scope_gradient_messages():
    """with dp_overlap (a collective's kernel may hold compute units under the backward pass) and a communicator
    attached, the stacks must not use the pair launches (all their workgroups have to be resident at once); the default tail
    message keeps them."""
    ps = make_store(1, [[2]], [32], seed=1)
    eng = _engine(2, 1, [[2]], [32], ps)
    c = eng.codecs[0]
    stack = c.enc_stages[0][0]
    if not eng.fused_pairs:
        pytest.skip("pair launches are off in this process (ranks share a device)")
    assert c._pair_ok(stack)
    eng.dp_overlap, eng._dp_comm_attached = True, True
    assert not c._pair_ok(stack)
    eng.dp_overlap = False
    assert c._pair_ok(stack)


def test_engine_step_on_nan_poisoned_lds_equals_the_plain_step():
    """Every launch of a whole train step (the config-3 joint step and a follower step: fused quantizer / chain epilogues, pair launches,
    batched weight gradients - the call patterns the per-kernel tests do not all reach) preceded by a launch that leaves NaN in the LDS
    of every CU (tests/test_kernels_gpu.py::_PoisonedLib): a kernel that reads LDS it never wrote would turn its output into NaN.  The
    gradients must equal the unpoisoned step's.  (Round 6: one of the things ruled out for the steps that differ when two processes
    share a GPU.)"""
    from tests.test_kernels_gpu import _PoisonedLib
    ps = make_store(2, [[2], [2]], [32, 32], lpc=True)
    x = dev(synth_frames(3).transpose(0, 2, 1))
    lx = dev(np.sort(np.random.default_rng(3).uniform(0.03, 3.1, (3, 16, 1)), axis=1).astype(np.float32))
    joint = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0, 10.0], c_ent=[0.3, 0.5], trainable=[True, True], lr=0.0, slot=1,
                 c_quan_lpc=10.0, train_lpc=True, quan_op=True)
    foll = dict(joint, c_quan=[0.0, 10.0], c_ent=[0.0, 0.3], trainable=[False, True], c_quan_lpc=0.0, train_lpc=False)
    for cfg in (joint, foll):
        got = {}
        for poisoned in (False, True):
            eng = _engine(3, 2, [[2], [2]], [32, 32], ps, res_scalar=2.0, scale_first=True, lpc=True)
            if poisoned:
                eng.lib = _PoisonedLib(eng.lib)
            eng.train_step(x, x, cfg, lpc_x=lx)
            torch.cuda.synchronize()
            got[poisoned] = (eng.grads.cpu().numpy().copy(), eng.decoded.cpu().numpy().copy())
        for a, b, what in zip(got[True], got[False], ("gradients", "decoded frames")):
            assert np.all(np.isfinite(a)), what
            assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max(), (what, float(np.abs(a - b).max()), float(np.abs(b).max()))


def test_live_gather_covers_every_word_a_step_reads():
    """The step's opening launch gathers only the regions of the kernel-ready images that the step's kernels are pointed at (recorded
    in the first run of a (flags, trainable pattern) pair: engine.wtp).  Proof that nothing else is read: everything the gather does not
    rewrite is NaN in the second run, and the gradients come out the same.  A new pattern / new flags gather everything again."""
    ps = make_store(2, [[2], [2]], [32, 32], lpc=True)
    eng = _engine(2, 2, [[2], [2]], [32, 32], ps, res_scalar=2.0, scale_first=True, lpc=True)
    if not eng.live_gather:
        pytest.skip("NSC_LIVE_GATHER=0")
    x = dev(synth_frames(2).transpose(0, 2, 1))
    lx = dev(np.sort(np.random.default_rng(3).uniform(0.03, 3.1, (2, 16, 1)), axis=1).astype(np.float32))
    joint = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0, 10.0], c_ent=[0.0, 0.0], trainable=[True, True], lr=0.0, slot=1,
                 c_quan_lpc=10.0, train_lpc=True, quan_op=True)
    foll = dict(joint, c_quan=[0.0, 10.0], c_ent=[0.0, 0.3], trainable=[False, True], c_quan_lpc=0.0, train_lpc=False)
    for cfg in (joint, foll):
        eng.train_step(x, x, cfg, lpc_x=lx)                      # full gather, records
        torch.cuda.synchronize()
        g1 = eng.grads.cpu().numpy().copy()
        key = [k for k in eng._live_tables if k[1] == tuple(cfg["trainable"])]
        assert len(key) == 1
        words = int(eng._live_tables[key[0]][0][:, 1].sum().item())
        assert 0 < words < eng.wt.numel()
        eng.wt.fill_(float("nan"))
        eng.train_step(x, x, cfg, lpc_x=lx)                      # gathers the recorded regions only
        torch.cuda.synchronize()
        g2 = eng.grads.cpu().numpy()
        assert np.all(np.isfinite(g2)) and np.abs(g2 - g1).max() <= 1e-6 * np.abs(g1).max(), (words, eng.wt.numel())
        print(f"live gather, trainable {cfg['trainable']}: {words} of {eng.wt.numel()} words")
        if cfg is joint:
            g_joint = g1
    assert words < int(eng._live_tables[[k for k in eng._live_tables if k[1] == (True, True)][0]][0][:, 1].sum().item())   # a frozen codec needs no backward images
    # a forward outside a step, on flags the last gather was not recorded for, must not trust the images (they are partly NaN here)
    eng.train_step(x, x, joint, lpc_x=lx)
    eng.images_valid = True            # (lr = 0: the parameters are the ones the step's gather read; Adam's launch had cleared the flag)
    assert eng.images_valid
    eng.split_fwd = False
    assert not eng.images_valid
    d = eng.forward(x, 1.0, True, lpc_x=lx)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(d).all())
    # ... and a step on the new flags gathers everything again (and records its own table)
    eng.wt.fill_(float("nan"))
    eng.train_step(x, x, joint, lpc_x=lx)
    torch.cuda.synchronize()
    g3 = eng.grads.cpu().numpy()
    assert np.all(np.isfinite(g3)) and np.abs(g3 - g_joint).max() <= 1e-3 * np.abs(g_joint).max()      # (exact against split forward)
    assert len(eng._live_tables) == 3


def test_tail_stream_is_off_under_per_scope_gradient_messages():
    """the second stream at the tail of the step (the convs' deferred weight gradients beside the blocks') is for the
    step that overlaps nothing else; with per-scope messages under the backward pass and a communicator attached the flushes stay on
    one stream.  And the two-stream tail gives the one-stream gradients (same kernels, same order per tensor)."""
    ps = make_store(1, [[2]], [32], seed=1)
    eng = _engine(2, 1, [[2]], [32], ps)
    assert eng._tail_two_streams() == eng.tail_overlap
    eng.dp_overlap, eng._dp_comm_attached = True, True
    assert not eng._tail_two_streams()
    eng.dp_overlap = eng._dp_comm_attached = False
    x = dev(synth_frames(2).transpose(0, 2, 1))
    cfg = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0], c_ent=[0.3], trainable=[True], lr=0.0, slot=1)
    got = {}
    for two in (True, False):
        eng.tail_overlap = two
        eng.train_step(x, x, cfg)
        torch.cuda.synchronize()
        got[two] = eng.grads.cpu().numpy().copy()
    assert np.abs(got[True] - got[False]).max() <= 1e-6 * np.abs(got[False]).max()


def test_steps_are_bit_stable_when_two_processes_share_the_gpu(tmp_path):
    """Round 6: with packed fp32 vector instructions in the library, 1-2 % of the engine's steps came out different in a few entries
    whenever a second process shared the GPU (the low half of a packed result, lanes 48..63 - profiles/r06_gpu_sharing_signature.txt);
    the library is built without them (csrc/Makefile).  The guard: two independent single-GPU processes side by side, 100 repetitions
    of the config-3 step per (tail stream, message layout) pair each, every repetition compared with the first - no difference beyond
    float-atomics noise (tools/dp_race_stress.py; the shipped flags measured 0 in 3 000 shared steps, the old ones 38)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSC_STRESS_SOLO="1", NSC_STRESS_RESTORE="lr0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "dp_race_stress.py"), "100"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
        lines = [l for l in out.splitlines() if l.startswith("restore=")]
        assert len(lines) == 4, out[-3000:]
        for l in lines:
            assert ": 0 glitches in 99 repetitions" in l, l


def test_bench_line_contract():
    """`python bench.py` (the command the driver runs) prints ONE JSON line with the contract's keys; the live roofline record carries
    the kernel's average launch time corrected by the event bracket's own overhead (measured in the same run) and stays below the peak."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-infer"],
                         capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"].startswith("f32") and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - 128 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]              # frames/s of the whole job
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "avg_launch_us_event_bracket",
              "event_bracket_overhead_us"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.05 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 <= r["event_bracket_overhead_us"] <= 10.0
    assert abs(r["avg_launch_us_event_bracket"] - r["avg_launch_us"] - r["event_bracket_overhead_us"]) < 0.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


@pytest.mark.parametrize("when", ["first", "timed"])
def test_bench_survives_a_pair_launch_timeout(when):
    """A neighbour wait of a pair launch that times out (another process or a collective holding compute units) must cost the run
    its pair launches, not its line: bench.py switches to one launch per block - before capture, or after the timed region and then
    times again - and says so."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSC_BENCH_FAKE_PAIR_TIMEOUT=when)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-infer",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["pair_launches"] is False and c["pair_launch_timeouts"] == 0 and "pair_launches_note" in c, c
    assert d["value"] > 0 and d["steps"] == 3
