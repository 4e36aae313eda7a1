"""GPU tests of the drop-in surface: the reference-named ops (autograd over the C ABI), the reference-named builder
methods, the CLI phases, and the framing helpers."""
import argparse
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT
from tests._util import BKD, assert_close, dev, make_store, relerr, synth_frames

pytestmark = pytest.mark.gpu


def _module():
    from nsc_amd.neural_speech_coding_module import neuralSpeechCodingModule
    m = neuralSpeechCodingModule.__new__(neuralSpeechCodingModule)
    m._bottleneck_kernel_and_dilation = list(BKD)
    return m


def test_op_surface_codec_matches_oracle_forward_and_gradients():
    """computational_graph_end2end_quan_on built from nn_core_operator ops == oracle codec (values + grads)."""
    from nsc_amd import loss_terms_and_measures as L
    from nsc_amd.scope import VariableStore, set_store
    B = 2
    ps = make_store(1, [[2]], [32])
    x = synth_frames(B)
    st = VariableStore(device="cuda")
    set_store(st)
    try:
        m = _module()
        xd = dev(x)
        # first pass creates the variables; then overwrite them with the oracle's values and re-trace
        m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
        assert list(st.vars.keys()) == list(ps.params.keys())          # same names, same creation order as TF
        with torch.no_grad():
            for k, v in st.vars.items():
                v.copy_(torch.tensor(np.asarray(ps.params[k], np.float32).reshape(tuple(v.shape)), device="cuda"))
        st.begin_pass()
        p, _, _, code0, decoded, alpha, bins, _ = m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
        tgt = xd[:, :, 0].contiguous()
        time_loss, freq_loss = L.mse_loss(decoded, tgt), L.mfcc_loss(decoded, tgt)
        loss = (60.0 * time_loss + 10.0 * freq_loss + 10.0 * L.quan_loss(p)).sum() + B * 0.4 * L.entropy_coding_loss(p)
        loss.backward()
        tp = OT.TorchParams(ps)
        outs, dec = OT.cascade_forward(torch.tensor(x), tp, BKD, [[2]], 1.0, True)
        ref = OT.total_loss_sum(dec, torch.tensor(x)[:, :, 0], [outs[0]["p"]], [60.0, 10.0, 10.0, 0.0], 0.4, "quan_last")
        ref.backward()
        assert_close(decoded.detach().cpu().numpy(), dec.detach().numpy(), what="surface decoded")
        assert_close(p.detach().cpu().numpy(), outs[0]["p"].detach().numpy(), what="surface p")
        assert abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
        for k, v in st.vars.items():
            g = tp.t[k].grad.numpy()
            assert relerr(v.grad.cpu().numpy().reshape(g.shape), g) < 5e-4, k
        # hard codes (the_share False) through the same surface
        st.begin_pass()
        _, _, _, _, dec_h, _, _, _ = m.computational_graph_end2end_quan_on(xd, False, 1.0, 32, "scope_1", [2])
        tp.reset()
        oh = OT.codec_forward(torch.tensor(x), tp, "scope_1", BKD, [2], 1.0, False)
        assert_close(dec_h.detach().cpu().numpy(), oh["decoded"].detach().numpy(), what="surface hard decoded")
    finally:
        set_store(None)


@pytest.mark.parametrize("cin,wide,dil,flat,T", [(100, 100, 1, False, 96), (100, 100, 2, True, 130), (50, 50, 2, False, 64),
                                                (1, 100, 1, False, 77), (1, 50, 2, True, 64), (24, 24, 1, False, 40)])
def test_fused_gated_bottleneck_matches_the_composed_ops_and_the_oracle(cin, wide, dil, flat, T):
    """nn_core_operator.gated_bottleneck as ONE fused call per direction (ops.BlockFn) against the same function composed op by op
    (FUSED_BLOCKS False) and against the float64 oracle: output, input gradient, all eight parameter gradients; same variable names
    in the same creation order."""
    from nsc_amd import nn_core_operator as nn
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    rng = np.random.default_rng(cin + 7 * dil + T)
    B = 3
    x_np = rng.standard_normal((B, T, cin)).astype(np.float32)
    w_np = rng.standard_normal((B, T, wide)).astype(np.float32)      # weights of the scalar the gradients are taken of
    res = {}
    try:
        for fused in (True, False):
            nn.FUSED_BLOCKS = fused
            st = VariableStore(device="cuda", seed=5)
            set_store(st)
            x = dev(x_np).requires_grad_(True)
            with variable_scope("s"):
                y = nn.gated_bottleneck(x, wide, 20, 9, 9, dil, flat)
            (y * dev(w_np)).sum().backward()
            res[fused] = (y.detach().cpu().numpy(), x.grad.cpu().numpy(), {k: v.grad.cpu().numpy().copy() for k, v in st.vars.items()},
                          {k: v.detach().cpu().numpy().copy() for k, v in st.vars.items()})
    finally:
        nn.FUSED_BLOCKS = True
        set_store(None)
    assert list(res[True][2].keys()) == list(res[False][2].keys()) == [
        f"s/conv1d{sfx}/{n}" for sfx in ("", "_1", "_2", "_3") for n in ("kernel", "bias")]
    for k in res[True][3]:
        assert np.array_equal(res[True][3][k], res[False][3][k]), k        # same initial values either way
    # float64 oracle on the same parameters
    tp_t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in res[True][3].items()}

    class _TP:
        t = tp_t
        n = 0

        def conv(self, scope):
            nm = "s/conv1d" + ("" if self.n == 0 else f"_{self.n}")
            self.n += 1
            return self.t[nm + "/kernel"], self.t[nm + "/bias"]
    xo = torch.tensor(x_np, dtype=torch.float64, requires_grad=True)
    yo = OT.gated_bottleneck(xo, _TP(), "s", dil, flat)
    (yo * torch.tensor(w_np, dtype=torch.float64)).sum().backward()
    for fused in (True, False):
        yv, dxv, gr, _ = res[fused]
        assert_close(yv, yo.detach().numpy(), what=f"block out fused={fused}")
        assert_close(dxv, xo.grad.numpy(), what=f"block dx fused={fused}")
        for k, g in gr.items():
            assert relerr(g.reshape(-1), tp_t[k].grad.numpy().reshape(-1)) < 2e-5, (fused, k)


def test_surface_ops_chain_without_transposes_and_accept_channels_last_memory(monkeypatch):
    """Between surface ops a tensor is [B,C,T] memory behind a [B,T,C] view: a chain conv -> lrelu -> block -> shuffle -> multiply
    launches nsc_transpose_last2 once for the channels_last input and once for the channels_last gradient torch hands to the last op,
    and nowhere in between; a real
    channels_last tensor produced by torch in the middle of the chain is still taken (transposed on the way in)."""
    from nsc_amd import _lib, nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    lib = _lib.load()
    calls = [0]
    real = lib.nsc_transpose_last2

    class _Counting:
        def __getattr__(self, name):
            f = getattr(lib, name)
            if name != "nsc_transpose_last2":
                return f

            def g(*a):
                calls[0] += 1
                return real(*a)
            return g
    monkeypatch.setattr(ops, "_lib_", lambda: _Counting())
    rng = np.random.default_rng(3)
    x_np = rng.standard_normal((2, 64, 6)).astype(np.float32)
    set_store(VariableStore(device="cuda", seed=1))
    try:
        x = dev(x_np).requires_grad_(True)
        with variable_scope("s"):
            h = nn.activation_func(nn.conv1d(x, 100, 9, activation=None))
            h = nn.gated_bottleneck(h, 100, 20, 9, 9, 2, False)
            up = ops.ShuffleFn.apply(h)                                   # [2,128,50]
            out = ops.MulFn.apply(up, up)
        assert out.shape == (2, 128, 50) and not out.is_contiguous() and out.transpose(1, 2).is_contiguous()
        assert calls[0] == 1
        out.sum().backward()
        assert calls[0] == 2 and x.grad.shape == x.shape
        # the shuffle itself against the reference's reshape / transpose (neural_speech_coding_module.py:158-167)
        assert np.array_equal(up.detach().cpu().numpy(), OT.subpixel_shuffle(h.detach().cpu(), 2).numpy())
        # channels_last memory in the middle (a torch op materialises [B,T,C]): same values as the lazy route
        with variable_scope("t"):
            a = nn.conv1d(h, 10, 3)
        set_store(VariableStore(device="cuda", seed=1))
        with variable_scope("s"):
            h2 = nn.activation_func(nn.conv1d(x, 100, 9, activation=None))
            h2 = nn.gated_bottleneck(h2, 100, 20, 9, 9, 2, False)
        with variable_scope("t"):
            b = nn.conv1d(h2.contiguous() * 1.0, 10, 3)
        assert b.is_contiguous() is False and np.array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy())
    finally:
        set_store(None)


def test_surface_rejects_cpu_tensors_and_bad_shapes():
    from nsc_amd import _lib, nn_core_operator as nn
    from nsc_amd.scope import VariableStore, set_store
    set_store(VariableStore(device="cuda"))
    try:
        with pytest.raises(_lib.NscError):
            nn.conv1d(torch.zeros(1, 8, 2), 4, 3)
        with pytest.raises(ValueError):
            nn.scalar_softmax_quantization(torch.zeros(1, 8, 1, device="cuda"), -1.0, torch.zeros(4, device="cuda"), 1.0, True, 8, 5)
        with pytest.raises(ValueError):
            nn.conv1d(torch.zeros(1, 8, 2, device="cuda"), 4, 3, padding='VALID')
    finally:
        set_store(None)


def _args(tmp, **kw):
    base = dict(learning_rate_tanh=2e-4, learning_rate_greedy_followers="2e-5 2e-6", epoch_tanh=2,
                epoch_greedy_followers="1 1", from_where_step=2, batch_size=8, num_resnets=2, training_mode="4",
                base_model_id="", suffix="end2endcascade", window_size=512, bottleneck_kernel_and_dilation="9 9 100 20 1 2",
                is_cq=0, the_strides="2 2", save_unique_mark="", coeff_term="60 10 10 0", res_scalar=1.0, pretrain_step=1,
                target_entropy=2.2, num_bins_for_follower="32 32", lpc_domain=False, data_root=None,
                max_batches_per_epoch=2, out_root=str(tmp), model_id="1234567", seed=1)
    base.update(kw)
    return argparse.Namespace(**base)


def test_cli_phases_cascaded_then_finetune_then_feedforward(tmp_path):
    """README flow of the reference: mode 4 (codec 1 pretrain+quan, follower), mode 5 (joint finetune), mode 0."""
    from nsc_amd.cmrl import CMRL
    a = _args(tmp_path, the_strides="2")
    m = CMRL(a)
    m.model("cascaded", a)
    ck = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "check" / "*.npz")))
    assert ck == ["model_bnn_ac_1234567_.ckpt.npz", "model_bnn_ac_1234567_follower_1end2endcascade.ckpt.npz"]
    z1 = np.load(tmp_path / "check" / ck[0])
    z2 = np.load(tmp_path / "check" / ck[1])
    k = "scope_1|conv1d_3|kernel"
    assert np.array_equal(z1[k], z2[k])                       # codec 1 is frozen during the follower phase
    assert "scope_2|conv1d|kernel" in z2.files and "scope_2|conv1d|kernel" not in z1.files
    journal = open(glob.glob(str(tmp_path / "doc" / "*_journal.txt"))[0]).read()
    assert journal.count("Epoch") == 3                        # 2 epochs of codec 1 + 1 follower epoch
    import re                                                 # the reference's line format (nsc_module:520-530)
    assert re.search(r"Epoch   0: SNR: [-\d.]+ dB Si-SNR: [-\d.]+ dB STOI: +nan PESQ: +nan _quan_loss: [\d.]+tau: [-\d.]+   "
                     r"fully_entropy: [\d.]+ \n", journal), journal
    bins_art = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "bins1234567*.npy")))
    assert bins_art == ["bins12345670.npy", "bins12345671.npy"]   # one per epoch that ran the quantizer (nsc_module:740)
    assert np.load(tmp_path / bins_art[-1]).shape == (32,)
    a5 = _args(tmp_path, the_strides="2", training_mode="5", base_model_id="1234567")
    m5 = CMRL(a5)
    m5.model("finetune", a5)
    z3 = np.load(tmp_path / "check" / "model_bnn_ac_1234567_finetune_2end2endcascade.ckpt.npz")
    assert not np.array_equal(z3[k], z2[k])                   # joint phase trains codec 1 again
    assert np.all(np.isfinite(z3[k]))
    a0 = _args(tmp_path, the_strides="2", training_mode="0", base_model_id="1234567")
    rng = np.random.default_rng(3)
    utts = [(0.03 * rng.standard_normal(n)).astype(np.float32) for n in (4000, 993, 513, 512)]
    m0 = CMRL(a0)
    outs = m0._feedforward(2, utterances=utts)
    assert [len(o) for o in outs] == [512 + 480 * 7, 992, 512, 0]      # frame counts of utilities.py:26
    assert all(np.all(np.isfinite(o)) for o in outs)
    journal = open(glob.glob(str(tmp_path / "doc" / "*_journal.txt"))[0]).read()
    assert journal.count("Test Utterance") == 3 and "PESQ-WB:" in journal


def test_cli_lpc_collaborative_quantisation_phase(tmp_path):
    """is_cq=1, LPC-residual domain (constants.is_pure_time_domain=False in the reference): one_ae_lpc + finetune_lpc."""
    from nsc_amd.cmrl import CMRL
    a = _args(tmp_path, the_strides="2 2", is_cq=1, lpc_domain=True, num_resnets=2, epoch_tanh=2)
    m = CMRL(a)
    m.model("cascaded", a)
    z = np.load(tmp_path / "check" / "model_bnn_ac_1234567_follower_1end2endcascade.ckpt.npz")
    assert "lpc_quan|bins" in z.files and z["lpc_quan|bins"].shape == (256,)
    a5 = _args(tmp_path, the_strides="2 2", is_cq=1, lpc_domain=True, training_mode="5", base_model_id="1234567")
    m5 = CMRL(a5)
    eng = m5._finetuning(2)
    assert eng.lpc and eng.scale_first and bool(torch.isfinite(eng.params).all())
    # the LSF quantizer is trainable under is_cq (its bins moved)
    z2 = np.load(tmp_path / "check" / "model_bnn_ac_1234567_finetune_2end2endcascade.ckpt.npz")
    assert not np.array_equal(z2["lpc_quan|bins"], z["lpc_quan|bins"])


def test_framing_on_gpu_roundtrip():
    from nsc_amd import utilities as U
    rng = np.random.default_rng(5)
    utt = rng.standard_normal(16000).astype(np.float32)
    fr = U.frames_on_gpu(torch.tensor(utt, device="cuda"), post_window=True)
    assert np.array_equal(fr.cpu().numpy(), O.utterance_to_segment(utt.astype(np.float64), True).astype(np.float32))
    ola = U.overlap_add_on_gpu(fr)
    ref = O.overlap_add(fr.cpu().numpy().astype(np.float64))
    assert_close(ola.cpu().numpy(), ref, tol=1e-6, what="overlap-add")
    # interior samples are reconstructed exactly where the Hann halves sum to one (mid windows)
    n = ola.numel()
    inner = slice(600, n - 600)
    assert np.max(np.abs(ola.cpu().numpy()[inner] - utt[:n][inner])) < 1e-5


def _cli(tmp, out, batch, nproc, data, extra=()):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--learning_rate_tanh", "2e-4", "--learning_rate_greedy_followers", "2e-5 2e-6", "--epoch_tanh", "3",
             "--epoch_greedy_followers", "1 1", "--from_where_step", "2", "--batch_size", str(batch), "--num_resnets", "1",
             "--training_mode", "1", "--base_model_id", "", "--suffix", "dp", "--window_size", "512",
             "--bottleneck_kernel_and_dilation", "9 9 100 20 1 2", "--is_cq", "0", "--the_strides", "2", "--save_unique_mark", "",
             "--coeff_term", "60 10 10 0.3", "--res_scalar", "1.0", "--pretrain_step", "1", "--target_entropy", "2.2",
             "--num_bins_for_follower", "32", "--data_root", str(data), "--max_batches_per_epoch", "3", "--out_root", str(out),
             "--model_id", "7654321", "--seed", "3", "--dump_rows", "1", *extra]
    env = dict(os.environ, NSC_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")     # the ranks share the test box's one GPU
    if nproc == 1:
        cmd = [sys.executable, os.path.join(root, "main.py"), *flags]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
               "127.0.0.1", "--master-port", "29641", os.path.join(root, "main.py"), *flags]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp))
    assert r.returncode == 0, r.stderr[-3000:]
    return open(glob.glob(os.path.join(str(out), "doc", "*_journal.txt"))[0]).read()


def test_cli_two_ranks_shard_the_data_file_and_match_one_process(tmp_path):
    """main.py mode 1 over a --data_root file: 2 ranks x batch 2 == 1 process x batch 4.  The ranks feed DISJOINT rows (each
    step's global batch is split between them, identical seeded row order on both), gradients are summed and the entropy
    histogram is that of the global batch - so the journal (SNR / quan loss / tau / entropy per epoch) is the single-process
    one up to fp32 reduction order.  Reference loop: neural_speech_coding_module.py:115-121, 424-549."""
    rng = np.random.default_rng(11)
    win = np.concatenate([np.hanning(63)[:32], np.ones(448), np.hanning(63)[31:]])
    data = (np.clip(0.03 * rng.standard_normal((30, 512)), -1, 1) * win).astype(np.float32)
    np.save(tmp_path / "frames.npy", data)
    (tmp_path / "one").mkdir(); (tmp_path / "two").mkdir()
    j1 = _cli(tmp_path, tmp_path / "one", 4, 1, tmp_path / "frames.npy")
    j2 = _cli(tmp_path, tmp_path / "two", 2, 2, tmp_path / "frames.npy")
    rows1 = np.load(tmp_path / "one" / "rows_rank0.npy")
    ra, rb = np.load(tmp_path / "two" / "rows_rank0.npy"), np.load(tmp_path / "two" / "rows_rank1.npy")
    assert rows1.shape == (9, 4) and ra.shape == rb.shape == (9, 2)            # 3 epochs x 3 steps
    assert np.array_equal(rows1, np.concatenate([ra, rb], axis=1))            # same global batches, split in rank order
    assert all(not set(a.tolist()) & set(b.tolist()) for a, b in zip(ra, rb))  # disjoint rows in every step
    import re
    num = lambda j: [float(v) for v in re.findall(r"(?<![\w.])-?\d+\.\d+(?:e-?\d+)?", j.split("\n\n", 1)[1])]
    a, b = np.array(num(j1)), np.array(num(j2))
    assert a.shape == b.shape and a.size >= 3 * 5, (j1, j2)          # SNR, Si-SNR, quan loss, tau, entropy per epoch
    assert np.allclose(a, b, rtol=2e-3, atol=2e-4), (j1, j2)
    # --local_entropy: the run still trains (finite journal), but the entropy term no longer sees the global histogram
    (tmp_path / "loc").mkdir()
    j3 = _cli(tmp_path, tmp_path / "loc", 2, 2, tmp_path / "frames.npy", extra=("--local_entropy", "1"))
    c = np.array(num(j3))
    assert c.shape == a.shape and np.all(np.isfinite(c[~np.isnan(a)]))
    # --tf_checkpoint: every checkpoint also in TensorFlow's V2 format under the reference's prefix, same variables, same values
    (tmp_path / "tf").mkdir()
    _cli(tmp_path, tmp_path / "tf", 4, 1, tmp_path / "frames.npy", extra=("--tf_checkpoint", "1"))
    from nsc_amd.tf_checkpoint import read_checkpoint
    idx = glob.glob(os.path.join(str(tmp_path / "tf"), "check", "model_bnn_ac_7654321_*.ckpt.index"))
    assert idx, os.listdir(tmp_path / "tf" / "check")
    for f in idx:
        named = read_checkpoint(f[:-len(".index")])
        with np.load(f[:-len(".index")] + ".npz") as z:
            assert set(named) == {k.replace("|", "/") for k in z.files} and len(named) >= 70
            for k in z.files:
                assert np.array_equal(named[k.replace("|", "/")], z[k]), k


@pytest.mark.parametrize("key,strides", [("s2", [2]), ("s22", [2, 2])])
def test_product_lpc_graph_builder_matches_reference(key, strides):
    """The PRODUCT's computational_graph_end2end_quan_on_lpc (nsc_amd/neural_speech_coding_module.py, the op-surface graph
    of nsc_module:297-335) against what the reference's own builder returned (fixture cg_*_lpc_*): the 7-tuple, soft / hard
    / unquantised decoded frames, the first frame's code and the soft assignment."""
    from nsc_amd.scope import VariableStore, set_store
    from tests._replay import FX, named_store
    x = FX["cg_x"]
    ps = named_store(1, [strides], [32])
    st = VariableStore(device="cuda")
    set_store(st)
    try:
        m = _module()
        xd = dev(x)
        r = m.computational_graph_end2end_quan_on_lpc(xd, None, True, 1.0, 32, "scope_1", strides)
        assert len(r) == 7                                                  # the _lpc variant returns 7 values (:335)
        assert list(st.vars.keys()) == list(ps.params.keys())
        with torch.no_grad():
            for k, v in st.vars.items():
                v.copy_(torch.tensor(np.asarray(ps.params[k], np.float32).reshape(tuple(v.shape)), device="cuda"))
        for sh in (True, False):
            st.begin_pass()
            p, _, _, code0, dec, alpha, bins = m.computational_graph_end2end_quan_on_lpc(xd, None, sh, 1.0, 32, "scope_1", strides)
            k = f"cg_{key}_lpc_{'soft' if sh else 'hard'}"
            assert_close(code0.detach().cpu().numpy(), FX[k + "_code0"], what=k + " code", atol=1e-6)
            assert_close(dec.detach().cpu().numpy(), FX[k + "_dec"], what=k + " decoded")
            if sh:
                assert_close(p.detach().cpu().numpy(), FX[k + "_p"], what=k + " p", atol=1e-6)
        st.begin_pass()
        dec0 = m.computational_graph_end2end_quan_on_lpc(xd, None, True, 0.0, 32, "scope_1", strides)[4]
        assert_close(dec0.detach().cpu().numpy(), FX[f"cg_{key}_lpc_noquan_dec"], what="lpc graph, is_quan_on = 0")
    finally:
        set_store(None)


# ---- round 6: what the surface derives from the parameters (one image gather per pass), the zero pool, deferred weight gradients ----
def _surface_step(st, m, xd, tgt, B, zero=True):
    from nsc_amd import loss_terms_and_measures as L
    st.begin_pass()
    if zero:
        for v in st.vars.values():
            v.grad = None
    p, _, _, _, decoded, _, _, _ = m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
    loss = (60.0 * L.mse_loss(decoded, tgt) + 10.0 * L.mfcc_loss(decoded, tgt) + 10.0 * L.quan_loss(p)).sum() + B * 0.4 * L.entropy_coding_loss(p)
    loss.backward()
    return loss.detach()             # (a tensor: the step may be under graph capture)


def _grads(st):
    torch.cuda.synchronize()
    return {k: v.grad.detach().cpu().numpy().copy() for k, v in st.vars.items()}


def _max_rel(a, b):
    return max(relerr(a[k], b[k]) for k in a)


def test_deferred_weight_gradients_equal_the_per_op_launches(monkeypatch):
    """ops.DEFER_WGRAD: the batched launches at the end of the pass give the gradients of the per-op launches - with .grad unset, with a
    .grad already there (accumulation over two passes), and through torch.autograd.grad."""
    from nsc_amd import ops
    from nsc_amd.scope import VariableStore, set_store
    B = 4
    xd = dev(synth_frames(B))
    tgt = xd[:, :, 0].contiguous()
    res = {}
    for defer in (False, True):
        monkeypatch.setattr(ops, "DEFER_WGRAD", defer)
        st = VariableStore(device="cuda", seed=11)
        set_store(st)
        try:
            m = _module()
            l1 = float(_surface_step(st, m, xd, tgt, B))
            g1 = _grads(st)
            l2 = float(_surface_step(st, m, xd, tgt, B, zero=False))          # .grad exists: autograd adds into it
            g2 = _grads(st)
            st.begin_pass()
            p, _, _, _, decoded, _, _, _ = m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
            from nsc_amd import loss_terms_and_measures as L
            loss = (60.0 * L.mse_loss(decoded, tgt) + 10.0 * L.quan_loss(p)).sum()
            names = list(st.vars)
            before = {k: st.vars[k].grad.detach().clone() for k in names}
            ga = torch.autograd.grad(loss, [st.vars[k] for k in names], allow_unused=True)
            torch.cuda.synchronize()
            for k in names:                                            # .grad itself is left alone by autograd.grad
                assert torch.equal(st.vars[k].grad, before[k]), k
            res[defer] = (l1, g1, l2, g2, {k: (g.cpu().numpy() if g is not None else None) for k, g in zip(names, ga)})
        finally:
            set_store(None)
    (l1a, g1a, l2a, g2a, gaa), (l1b, g1b, l2b, g2b, gab) = res[False], res[True]
    assert abs(l1a - l1b) <= 1e-6 * abs(l1a) and abs(l2a - l2b) <= 1e-6 * abs(l2a)
    assert _max_rel(g1a, g1b) < 2e-5, _max_rel(g1a, g1b)
    assert _max_rel(g2a, g2b) < 2e-5
    for k in g1a:
        assert relerr(g2b[k], 2.0 * g1b[k]) < 2e-5, k                  # the second pass doubled every gradient
        assert (gaa[k] is None) == (gab[k] is None), k
        if gaa[k] is not None:
            assert relerr(gab[k], gaa[k]) < 2e-5, k
            assert np.abs(gab[k]).max() > 0 or np.abs(gaa[k]).max() == 0, k


def test_a_parameter_shared_by_two_blocks_gets_both_gradients(monkeypatch):
    """The same eight variables used by two fused blocks in one pass (begin_pass between the calls hands the variables out again): the
    deferred batch fills two zeroed views and autograd's sum of them ends up holding both contributions."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(5).standard_normal((2, 128, 100)).astype(np.float32)
    res = {}
    for defer in (False, True):
        monkeypatch.setattr(ops, "DEFER_WGRAD", defer)
        st = VariableStore(device="cuda", seed=3)
        set_store(st)
        try:
            x = dev(x_np)
            with variable_scope("s"):
                h = nn.gated_bottleneck(x, 100, 20, 9, 9, 1, False)
            st.begin_pass()
            with variable_scope("s"):
                h = nn.gated_bottleneck(h, 100, 20, 9, 9, 1, True)
            assert len(st.vars) == 8
            (h * h).sum().backward()
            res[defer] = _grads(st)
        finally:
            set_store(None)
    assert _max_rel(res[False], res[True]) < 2e-5, _max_rel(res[False], res[True])


def test_image_set_follows_the_parameters_between_passes_and_inside_a_captured_graph():
    """One gather per pass rebuilds every derived image: parameters changed between passes (in place, through .data, by an optimizer)
    are seen by the next pass, eagerly and by the REPLAY of a captured step (the gather is part of the graph; so are the zero pool's
    fill and the deferred batches - a replay gives the gradients of its own parameters, not an accumulation)."""
    from nsc_amd.scope import VariableStore, set_store
    B = 4
    xd = dev(synth_frames(B))
    tgt = xd[:, :, 0].contiguous()
    st = VariableStore(device="cuda", seed=21)
    set_store(st)
    try:
        m = _module()
        _surface_step(st, m, xd, tgt, B)
        _surface_step(st, m, xd, tgt, B)                                # second pass: the whole set in one launch
        iset, = st.image_sets.values()
        assert iset.buf is not None and iset.n_all == len(iset.items) >= 10
        la = float(_surface_step(st, m, xd, tgt, B))
        ga = _grads(st)
        saved = {k: v.detach().clone() for k, v in st.vars.items()}
        with torch.no_grad():
            for k, v in st.vars.items():
                if k.endswith("kernel"):
                    v.data.mul_(1.25)                                   # (.data: no version bump - the pass boundary is what counts)
        lb = float(_surface_step(st, m, xd, tgt, B))
        gb = _grads(st)
        assert abs(lb - la) > 1e-3 * abs(la) and _max_rel(ga, gb) > 1e-3
        # a fresh store holding the same values agrees: nothing stale survived the change
        st2 = VariableStore(device="cuda", seed=99)
        set_store(st2)
        m.computational_graph_end2end_quan_on(xd, True, 1.0, 32, "scope_1", [2])
        with torch.no_grad():
            for k, v in st2.vars.items():
                v.copy_(st.vars[k])
        lc = float(_surface_step(st2, m, xd, tgt, B))
        gc = _grads(st2)
        assert abs(lc - lb) <= 1e-6 * abs(lb) and _max_rel(gb, gc) < 1e-5
        # captured step: replay after restoring the old values gives the old gradients, after scaling again the new ones
        set_store(st)
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            _surface_step(st, m, xd, tgt, B)
        torch.cuda.current_stream().wait_stream(s_)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s_):
            _surface_step(st, m, xd, tgt, B)
        with torch.no_grad():
            for k, v in st.vars.items():
                v.copy_(saved[k])
        graph.replay()
        g_old = _grads(st)
        graph.replay()
        assert _max_rel(g_old, _grads(st)) < 1e-6                       # (float atomics in the slab-free kernels: not bit for bit)
        assert _max_rel(ga, g_old) < 1e-5
        with torch.no_grad():
            for k, v in st.vars.items():
                if k.endswith("kernel"):
                    v.mul_(1.25)
        graph.replay()
        assert _max_rel(gb, _grads(st)) < 1e-5
    finally:
        set_store(None)


@pytest.mark.parametrize("T", [128, 126])
def test_fused_block_on_plain_tensors_outside_any_store(T):
    """ops.BlockFn on eight ordinary tensors (no VariableStore): the image pair comes from one gather over wherever they lie, or the
    pointer entry points take over - same values as the store route.  T = 126: no split-operand forward (T % 4 != 0), the exact
    kernels serve the block on both routes."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(8).standard_normal((2, T, 100)).astype(np.float32)
    st = VariableStore(device="cuda", seed=4)
    set_store(st)
    try:
        x = dev(x_np).requires_grad_(True)
        with variable_scope("s"):
            y = nn.gated_bottleneck(x, 100, 20, 9, 9, 2, False)
        (y * y).sum().backward()
        ref = (y.detach().cpu().numpy(), x.grad.cpu().numpy(), _grads(st))
        names = list(st.vars)
    finally:
        set_store(None)
    plain = [st.vars[k].detach().clone().requires_grad_(True) for k in names]     # separate allocations
    x2 = dev(x_np).requires_grad_(True)
    y2 = ops.BlockFn.apply(x2, *plain, 2, False)
    (y2 * y2).sum().backward()
    torch.cuda.synchronize()
    assert relerr(y2.detach().cpu().numpy(), ref[0]) < 1e-6
    assert relerr(x2.grad.cpu().numpy(), ref[1]) < 1e-5
    for k, t in zip(names, plain):
        assert relerr(t.grad.cpu().numpy(), ref[2][k]) < 2e-5, k


def test_two_losses_of_one_tensor_share_a_node_and_separate_backwards_still_work():
    from nsc_amd import loss_terms_and_measures as L, ops
    rng = np.random.default_rng(2)
    d_np, o_np = (0.1 * rng.standard_normal((4, 512))).astype(np.float32), (0.1 * rng.standard_normal((4, 512))).astype(np.float32)
    d, o = dev(d_np).requires_grad_(True), dev(o_np)
    t, f = L.mse_loss(d, o), L.mfcc_loss(d, o)
    assert t.grad_fn is f.grad_fn                                       # one ReconLossFn node
    (60.0 * t + 10.0 * f).sum().backward()
    g_joint = d.grad.clone()
    d.grad = None
    L.mse_loss(d, o).sum().mul(60.0).backward()                         # a node that has run is not handed out again
    L.mfcc_loss(d, o).sum().mul(10.0).backward()
    assert relerr(d.grad.cpu().numpy(), g_joint.cpu().numpy()) < 1e-5
    dd = torch.tensor(d_np, dtype=torch.float64, requires_grad=True)
    (60.0 * OT.mse_loss(dd, torch.tensor(o_np, dtype=torch.float64)) + 10.0 * OT.mfcc_loss(dd, torch.tensor(o_np, dtype=torch.float64))).sum().backward()
    assert relerr(g_joint.cpu().numpy(), dd.grad.numpy()) < 5e-4
    p = torch.softmax(dev(rng.standard_normal((4, 256, 32)).astype(np.float32)), -1).requires_grad_(True)
    q, e = L.quan_loss(p), L.entropy_coding_loss(p)
    assert q.grad_fn is e.grad_fn
    (q.sum() + 3.0 * e).backward()
    pp = p.detach().cpu().double().requires_grad_(True)
    (OT.quan_loss(pp).sum() + 3.0 * OT.entropy_coding_loss(pp)).backward()
    assert relerr(p.grad.cpu().numpy(), pp.grad.numpy()) < 1e-5


@pytest.mark.parametrize("C_,cin,flat", [(100, 100, True), (50, 50, False), (100, 1, True)])
def test_block_stack_equals_the_blocks_one_by_one(C_, cin, flat):
    """nn.gated_bottleneck_stack (one autograd node, the leaky-relu between the blocks differentiated inside the next block's
    data-gradient kernel) against the same blocks as separate gated_bottleneck calls: same variables, values and gradients."""
    from nsc_amd import nn_core_operator as nn
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(12).standard_normal((2, 192, cin)).astype(np.float32)
    w_np = np.random.default_rng(13).standard_normal((2, 192, C_)).astype(np.float32)
    res = {}
    for stacked in (True, False):
        st = VariableStore(device="cuda", seed=6)
        set_store(st)
        try:
            x = dev(x_np).requires_grad_(True)
            with variable_scope("s"):
                if stacked:
                    y = nn.gated_bottleneck_stack(x, C_, 20, 9, [1, 2, 1], is_last_flat=flat)
                else:
                    y = nn.gated_bottleneck(x, C_, 20, 9, 9, 1, False)
                    y = nn.gated_bottleneck(y, C_, 20, 9, 9, 2, False)
                    y = nn.gated_bottleneck(y, C_, 20, 9, 9, 1, flat)
            (y * dev(w_np)).sum().backward()
            res[stacked] = (list(st.vars), y.detach().cpu().numpy(), x.grad.cpu().numpy(), _grads(st))
        finally:
            set_store(None)
    (na, ya, dxa, ga), (nb, yb, dxb, gb) = res[True], res[False]
    assert na == nb and len(na) == 24
    assert np.array_equal(ya, yb)
    assert relerr(dxa, dxb) < 1e-6
    assert _max_rel(ga, gb) < 1e-5, _max_rel(ga, gb)


@pytest.mark.parametrize("C_", [100, 50])
def test_fused_up_sampling_equals_the_three_ops(C_):
    """nn.conv1d_depth_shuffle (ops.UpsampleFn: nsc_upsample_fwd / _bwd) against conv1d_depth -> shuffle composed op by op: same
    variables, values and gradients (neural_speech_coding_module.py:158-181)."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(21).standard_normal((3, 128, C_)).astype(np.float32)
    w_np = np.random.default_rng(22).standard_normal((3, 256, C_ // 2)).astype(np.float32)
    res = {}
    for fused in (True, False):
        st = VariableStore(device="cuda", seed=9)
        set_store(st)
        try:
            x = dev(x_np).requires_grad_(True)
            with variable_scope("s"):
                if fused:
                    y = nn.conv1d_depth_shuffle(x, C_, 9, activation='lrelu', stride=2)
                else:
                    y = ops.ShuffleFn.apply(nn.activation_func(nn.conv1d_depth(x, C_, 9, activation=None)))
            with torch.no_grad():
                st.vars["s/separable_conv1d/bias"].add_(0.01)            # (zero-initialised: give the bias path something to do)
            assert y.shape == (3, 256, C_ // 2)
            (y * dev(w_np)).sum().backward()
            res[fused] = (list(st.vars), y.detach().cpu().numpy(), x.grad.cpu().numpy(), _grads(st))
        finally:
            set_store(None)
    (na, ya, dxa, ga), (nb, yb, dxb, gb) = res[True], res[False]
    assert na == nb == ["s/separable_conv1d/depthwise_kernel", "s/separable_conv1d/pointwise_kernel", "s/separable_conv1d/bias"]
    assert relerr(ya, yb) < 1e-6 and relerr(dxa, dxb) < 1e-5
    assert _max_rel(ga, gb) < 2e-5, _max_rel(ga, gb)


def test_in_place_parameter_updates_are_seen_without_a_new_pass():
    """Store variables handed to ops.BlockFn directly, twice, with an optimizer-style in-place update in between and NO begin_pass():
    the image set notices the version counters of the block's tensors and gathers again."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x = dev(np.random.default_rng(31).standard_normal((2, 128, 100)).astype(np.float32))
    st = VariableStore(device="cuda", seed=14)
    set_store(st)
    try:
        with variable_scope("s"):
            nn.gated_bottleneck(x, 100, 20, 9, 9, 1, True)               # creates the eight variables (and registers the image pair)
        vs = list(st.vars.values())
        st.begin_pass()
        y0 = ops.BlockFn.apply(x, *vs, 1, True).detach().clone()         # pass 2: served from the consolidated buffer
        y0b = ops.BlockFn.apply(x, *vs, 1, True).detach().clone()
        assert torch.equal(y0, y0b)
        with torch.no_grad():
            for v in vs:
                v.mul_(1.5)                                              # what an optimizer step does: in place, version bump
        y1 = ops.BlockFn.apply(x, *vs, 1, True).detach().clone()         # same pass id
        ref = ops.BlockFn.apply(x, *[v.detach().clone() for v in vs], 1, True).detach()   # plain tensors: gathered per call
        torch.cuda.synchronize()
        assert relerr(y1.cpu().numpy(), ref.cpu().numpy()) < 1e-6
        assert relerr(y1.cpu().numpy(), y0.cpu().numpy()) > 1e-2
    finally:
        set_store(None)


def test_frozen_blocks_skip_their_weight_gradients_and_pass_the_data_gradient():
    """Convs and a stack whose variables do not require gradients (an earlier codec in a follower phase, cmrl.py:106-113): no
    weight-gradient job is queued for them, dx is what the trainable graph gives."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(41).standard_normal((2, 128, 100)).astype(np.float32)
    res = {}
    for frozen in (False, True):
        st = VariableStore(device="cuda", seed=17)
        set_store(st)
        try:
            x = dev(x_np).requires_grad_(True)

            def graph():
                with variable_scope("s"):
                    c = nn.conv1d(x, 100, 9, strides=2, activation='lrelu')          # (the split-operand stride-2 conv)
                    c = nn.conv1d(c, 100, 3, activation=None)                        # (a generic conv)
                    return nn.gated_bottleneck_stack(c, 100, 20, 9, [1, 2], is_last_flat=True)
            y = graph()
            if frozen:
                for v in st.vars.values():
                    v.requires_grad_(False)
                st.begin_pass()
                y = graph()
            n0 = len(ops._PENDING)
            (y * y).sum().backward()
            torch.cuda.synchronize()
            assert len(ops._PENDING) == n0
            res[frozen] = x.grad.cpu().numpy()
            assert all((v.grad is None) == frozen for v in st.vars.values())
        finally:
            set_store(None)
    assert relerr(res[True], res[False]) < 1e-6


def test_surface_blocks_on_the_exact_instruction_agree_with_the_split_operands(monkeypatch):
    """NSC_BLOCK_ARITH=exact (ops.SPLIT_ARITH False): the surface's blocks and stride-2 conv on the fp32 matrix instruction - the same
    values and gradients as the default split-operand kernels to fp32-class error, in the same store (the image set keeps both kinds)."""
    from nsc_amd import nn_core_operator as nn, ops
    from nsc_amd.scope import VariableStore, set_store, variable_scope
    x_np = np.random.default_rng(51).standard_normal((2, 256, 100)).astype(np.float32)
    w_np = np.random.default_rng(52).standard_normal((2, 128, 100)).astype(np.float32)
    st = VariableStore(device="cuda", seed=23)
    set_store(st)
    try:
        res = {}
        for split in (True, False, True):
            monkeypatch.setattr(ops, "SPLIT_ARITH", split)
            st.begin_pass()
            for v in st.vars.values():
                v.grad = None
            x = dev(x_np).requires_grad_(True)
            with variable_scope("s"):
                c = nn.conv1d(x, 100, 9, strides=2, activation='lrelu')
                y = nn.gated_bottleneck_stack(c, 100, 20, 9, [1, 2], is_last_flat=False)
            (y * dev(w_np)).sum().backward()
            cur = (y.detach().cpu().numpy(), x.grad.cpu().numpy(), _grads(st))
            if split in res:                                             # the second split pass: the first one's results again
                assert np.array_equal(cur[0], res[True][0])
            res[split] = cur
        ys, dxs, gs = res[True]
        ye, dxe, ge = res[False]
        assert 0 < relerr(ys, ye) < 2e-5 and relerr(dxs, dxe) < 5e-5      # (different arithmetic, fp32-class agreement)
        assert _max_rel(gs, ge) < 1e-4, _max_rel(gs, ge)
    finally:
        set_store(None)


def test_captured_surface_step_trains_like_the_eager_one():
    """nsc_amd.graph.capture_step: three SGD steps (in-place parameter updates between replays, new frames copied into the static input)
    through the replayed graph end where three eager steps end."""
    from nsc_amd.graph import capture_step
    from nsc_amd.scope import VariableStore, set_store
    B = 4
    frames = [dev(synth_frames(B + 8)[k:k + B]) for k in (0, 4, 8)]
    res = {}
    for mode in ("eager", "graph"):
        st = VariableStore(device="cuda", seed=33)
        set_store(st)
        try:
            m = _module()
            xd = frames[0].clone()
            tgt = torch.empty((B, 512), device="cuda")

            def step():
                tgt.copy_(xd[:, :, 0])
                _surface_step(st, m, xd, tgt, B)
            step()                                                       # creates the variables
            run = capture_step(step) if mode == "graph" else step
            for f in frames:
                xd.copy_(f)
                run()
                with torch.no_grad():
                    for v in st.vars.values():
                        v.add_(v.grad, alpha=-1e-3)
            torch.cuda.synchronize()
            res[mode] = {k: v.detach().cpu().numpy().copy() for k, v in st.vars.items()}
        finally:
            set_store(None)
    assert _max_rel(res["eager"], res["graph"]) < 1e-5, _max_rel(res["eager"], res["graph"])
