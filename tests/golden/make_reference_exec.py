#!/usr/bin/env python3
"""Generate tests/golden/reference_exec.npz by RUNNING the reference's own code (build container only).

    python tests/golden/make_reference_exec.py

The reference (``/root/reference``, read-only, never shipped) is imported on top of ``tf_shim.py`` (a lazy-graph
``tf`` stand-in that executes in float64) via ``ref_env.install()``.  Everything recorded below is the output of a
reference function / method call - this script holds no arithmetic of its own beyond seeded input generation.
Initial weights are name-seeded (``tf_shim.name_seeded_uniform``), so the fixture stores inputs, outputs,
losses, gradients and parameter snapshots (strided samples + sums), not 350 k-value weight sets.

Sections (key prefixes in the .npz):
  q_*    scalar_softmax_quantization          nn_core_operator.py:140-164
  gb_*   gated_bottleneck                     nn_core_operator.py:82-112
  ls_*   mse_loss / mfcc_loss / quan_loss / entropy_coding_loss      loss_terms_and_measures.py:77-84,151-183,257-267
  cg_*   computational_graph_end2end_quan_on[_lpc], strides [2] and [2,2]      neural_speech_coding_module.py:262-335
  ff4_*  all_modules_feedforward(4), each codec [2,2] (BASELINE config 4 forward)     cmrl.py:513-543
  td_*   one_ae -> _greedy_followers(1) -> _finetuning(2): per-step loss vectors, first-step gradients, tau
         trajectory, validation entropies, journal, checkpoints      nsc_module:891-939,424-549,657-758; cmrl.py:22-135,295-390
  lp_*   one_ae_lpc -> _greedy_followers_lpc(1) -> _finetuning_lpc(2)   nsc_module:989-1073,551-655; cmrl.py:137-293,392-511
  lu_*   lsf2poly_after_quan / lpc_analysis_get_residual / lpc_synthesizer_tr    lpc_utilities.py:28-77,137-156
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402

OUT = os.path.join(HERE, "reference_exec.npz")
BKD = [9, 9, 100, 20, 1, 2]
STRIDE = 31    # parameter / gradient tensors are stored as v.ravel()[::STRIDE] plus (sum, sum of squares)


def f32(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def synth_frames(n, seed, window):
    rng = np.random.default_rng(seed)
    return f32(np.clip(0.03 * rng.standard_normal((n, 512)), -1, 1) * window[None, :])


def sample(v):
    v = np.asarray(v, np.float64).ravel()
    return np.concatenate([[v.sum(), (v * v).sum()], v[::STRIDE]])


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        yield buf


def new_cmrl(E, **kw):
    obj = E.R.CMRL.__new__(E.R.CMRL)
    d = dict(_learning_rate_tanh=2e-4, _coeff_term=[60.0, 10.0, 10.0, 0.3], _pretrain_step=1, _target_entropy=2.2,
             _the_strides=[2], _res_scalar=1.0, _save_unique_mark="", _num_bins_for_follower=[32, 32, 32, 32],
             _epoch_tanh=3, _epoch_greedy_followers=[2, 2], _batch_size=2, _training_mode=3,
             _sep_val=["val_%d.wav" % i for i in range(10)], _sep_test=[], _rand_model_id="1234567",
             _base_model_id="1234567", _suffix="end2endcascade", _bottleneck_kernel_and_dilation=list(BKD),
             _window_size=512, _num_resnets=2, _from_where_step=2, _learning_rate_greedy_followers=[2e-4, 1e-4],
             _max_amp=E.C.max_amp_tr, _lpc_order=16, _is_cq=1, _tr_data_size=6)
    d.update(kw)
    for k, v in d.items():
        setattr(obj, k, v)
    return obj


# ------------------------------------------------------------------------------------------------
def run_ops(E, out):
    tf, N, L = E.tf, E.N, E.L
    v1 = tf.compat.v1
    rng = np.random.default_rng(11)
    # ---- quantizer: time-domain instance (nb=32, L=64) with exact ties, and the LSF instance (nb=256, L=16)
    code = f32(np.tanh(rng.standard_normal((3, 64, 1))))
    bins32 = f32(np.linspace(-1, 1, 32))
    code[0, 0, 0] = f32(0.5 * (bins32[3] + bins32[4]))     # may or may not tie after rounding: both sides recorded
    code[0, 1, 0] = bins32[7]
    code[0, 2, 0] = 0.0                                     # exact midpoint of the symmetric grid -> tie
    lsf = f32(np.sort(rng.uniform(0.03, 3.1, (3, 16, 1)), axis=1))
    out["q_code"], out["q_lsf"] = code, lsf
    for tag, x, nb, Lc, bins_init in (("c", code, 32, 64, np.linspace(-1, 1, 32)),
                                       ("l", lsf, 256, 16, np.asarray(E.C.lpc_coeff_lsf_bins))):
        for alpha0 in (-300.0, -20.0):
            with tf.Graph().as_default():
                c = v1.placeholder(tf.float32, (None, Lc, 1), "c")
                share = v1.placeholder(tf.bool, None, "share")
                qon = v1.placeholder(tf.float32, None, "qon")
                alpha = tf.Variable(alpha0, dtype=tf.float32, name="alpha")
                bins = tf.Variable(bins_init, dtype=tf.float32, name="bins")
                with quiet():
                    p, o = N.scalar_softmax_quantization(c, alpha, bins, qon, share, Lc, nb)
                ql, el = L.quan_loss(p), L.entropy_coding_loss(p)
                with v1.Session() as s:
                    s.run(v1.global_variables_initializer())
                    for sh in (True, False):
                        for qv in (1.0, 0.0, 0.5):
                            pv, ov, qlv, elv = s.run([p, o, ql, el], {c: x, share: sh, qon: qv})
                            k = f"q_{tag}_a{int(-alpha0)}_{'soft' if sh else 'hard'}_q{int(qv * 10)}"
                            out[k + "_out"] = ov
                            if sh and qv == 1.0:
                                out[k + "_p"], out[k + "_quan"], out[k + "_ent"] = pv, qlv, elv
    # ---- gated bottleneck: C=100 dil 1 / 2, flat and not; Cin = 1 broadcast residual
    for tag, cin, wide, dil, flat in (("a", 100, 100, 1, False), ("b", 100, 100, 2, True), ("c", 50, 50, 2, False),
                                      ("d", 1, 100, 1, False)):
        x = f32(0.5 * rng.standard_normal((2, 96, cin)))
        with tf.Graph().as_default():
            xi = v1.placeholder(tf.float32, (None, 96, cin), "x")
            with v1.variable_scope("gb_" + tag):
                y = N.gated_bottleneck(xi, wide_layer=wide, narrow_layer=20, non_dilated_neck_kernel_size=9,
                                       dilated_neck_kernel_size=9, dilation_rate=dil, is_last_flat=flat)
            with v1.Session() as s:
                s.run(v1.global_variables_initializer())
                # give the biases non-zero values (TF initialises them to 0; any value is a legal state)
                for v in v1.trainable_variables():
                    if v.var_name.endswith("/bias"):
                        v.assign_numpy(E.shim.name_seeded_uniform(v.var_name, v.trace.shape, 0.05))
                out[f"gb_{tag}_x"], out[f"gb_{tag}_y"] = x, s.run(y, {xi: x})
                out[f"gb_{tag}_cfg"] = np.array([cin, wide, dil, int(flat)])
    # ---- losses
    win = E.U.utterance_to_segment(np.ones(1200), False)[0]
    a = synth_frames(4, 21, win)
    b = f32(a + 0.01 * np.random.default_rng(22).standard_normal(a.shape))
    b[3] = a[3]                                            # identical pair: the 1e-7 floors decide the value
    pr = np.random.default_rng(23).random((4, 16, 32)) ** 4
    pr = f32(pr / pr.sum(-1, keepdims=True))
    with tf.Graph().as_default():
        d = v1.placeholder(tf.float32, (None, 512), "d")
        o = v1.placeholder(tf.float32, (None, 512), "o")
        p = v1.placeholder(tf.float32, (None, 16, 32), "p")
        with quiet():
            nodes = [L.mse_loss(d, o), L.mfcc_loss(d, o), L.quan_loss(p), L.entropy_coding_loss(p),
                     L.tf_stft(d)[1]]
        with v1.Session() as s:
            r = s.run(nodes, {d: b, o: a, p: pr})
    out["ls_dec"], out["ls_ori"], out["ls_p"] = b, a, pr
    out["ls_mse"], out["ls_mfcc"], out["ls_quan"], out["ls_ent"], out["ls_mag"] = r


def run_codec_graphs(E, out):
    tf = E.tf
    v1 = tf.compat.v1
    win = E.U.utterance_to_segment(np.ones(1200), False)[0]
    x = synth_frames(2, 31, win)[..., None]
    out["cg_x"] = x
    for key, strides in (("s2", [2]), ("s22", [2, 2])):
        for lpc in (False, True):
            obj = new_cmrl(E)
            with tf.Graph().as_default():
                xi = v1.placeholder(tf.float32, (None, 512, 1), "x")
                share = v1.placeholder(tf.bool, None, "share")
                qon = v1.placeholder(tf.float32, None, "qon")
                with quiet():
                    if lpc:
                        r = obj.computational_graph_end2end_quan_on_lpc(xi, None, share, qon, 32, "scope_1", strides)
                        assert len(r) == 7
                    else:
                        r = obj.computational_graph_end2end_quan_on(xi, share, qon, 32, "scope_1", strides)
                        assert len(r) == 8 and r[0] is r[7]
                names = [v.var_name for v in v1.trainable_variables()]
                shapes = [list(v.trace.shape) for v in v1.trainable_variables()]
                with v1.Session() as s:
                    s.run(v1.global_variables_initializer())
                    for sh in (True, False):
                        p, code0, dec = s.run([r[0], r[3], r[4]], {xi: x, share: sh, qon: 1.0})
                        k = f"cg_{key}_{'lpc' if lpc else 'td'}_{'soft' if sh else 'hard'}"
                        out[k + "_code0"], out[k + "_dec"] = code0, dec
                        if sh:
                            out[k + "_p"] = p
                    dec0 = s.run(r[4], {xi: x, share: True, qon: 0.0})
                    out[f"cg_{key}_{'lpc' if lpc else 'td'}_noquan_dec"] = dec0
                if not lpc:
                    out[f"cg_{key}_varnames"] = np.array(json.dumps([names, shapes]))


def run_feedforward4(E, out):
    """all_modules_feedforward: 4 codecs x [2,2] (BASELINE config 4) and 2 codecs x [2] (config 5), soft and hard."""
    tf = E.tf
    v1 = tf.compat.v1
    win = E.U.utterance_to_segment(np.ones(1200), False)[0]
    x = synth_frames(2, 41, win)[..., None]
    for tag, n, strides in (("ff4", 4, [2, 2]), ("ff2", 2, [2])):
        obj = new_cmrl(E, _the_strides=strides, _res_scalar=2.0, _num_resnets=n)
        with tf.Graph().as_default():
            with quiet():
                (xi, x_, lr, tau, share, qon, soft, enc, _, res, alpha, bins) = obj.all_modules_feedforward(n)
                ents = [E.L.entropy_coding_loss(p) for p in soft]
            with v1.Session() as s:
                s.run(v1.global_variables_initializer())
                for sh in (True, False):
                    outs, ev = s.run([res, ents], {xi: x, share: sh, qon: 1.0})
                    out[f"{tag}_{'soft' if sh else 'hard'}_yhat"] = np.stack(outs, 0)
                    out[f"{tag}_{'soft' if sh else 'hard'}_ent"] = np.array(ev)
        out[tag + "_x"] = x


# ------------------------------------------------------------------------------------------------
class PhaseRecorder:
    """Collects what the reference's training code does, through the shim's hooks."""

    def __init__(self, E):
        self.E = E
        self.steps = []
        self.first_grads = {}
        self.evals = []

    def hook(self, op, feeds, loss, grads):
        named = {}
        for k, v in feeds.items():
            if getattr(k, "name", None) in ("x", "x_", "lr", "tau", "is_quan_on", "lpc_x"):
                named[k.name] = v.numpy().copy()
            elif not hasattr(k, "dtype"):            # a fed non-placeholder tensor: the residual override
                named["res_x"] = v.numpy().copy()
        opt_id = self.E.shim.AdamOptimizer.OPTIMIZERS.index(op.opt)
        self.steps.append(dict(loss=loss, opt=opt_id, n_vars=len(op.var_list), **named))
        if op.opt.steps == 1:
            self.first_grads[opt_id] = {k: (None if g is None else sample(g)) for k, g in grads.items()}


def snapshot(path_npz):
    d = np.load(path_npz)
    return {k: sample(d[k]) for k in d.files}


def run_time_domain_phases(E, out):
    tf, M, U = E.tf, E.M, E.U
    win = U.utterance_to_segment(np.ones(1200), False)[0]
    tr = synth_frames(6, 51, win)
    vals = [f32(0.03 * np.random.default_rng(60 + i).standard_normal(1500)) for i in range(10)]
    out["td_tr_data"], out["td_val_utts"] = tr, np.stack(vals, 0)
    U.stoi = lambda *a, **k: 0.5
    U.pesq = lambda *a, **k: 1.0
    U.sf = type("sf", (), {"write": staticmethod(lambda *a, **k: None)})
    work = tempfile.mkdtemp(prefix="nsc_ref_td_")
    os.makedirs(os.path.join(work, "check"))
    os.makedirs(os.path.join(work, "doc"))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        random.seed(5)
        np.random.seed(5)
        obj = new_cmrl(E)
        obj._tr_data = tr.copy()
        obj._load_sig = lambda f: (vals[int(f.split("_")[1].split(".")[0])].copy(), 1.0)
        orig_eval = obj.end2end_eval
        evals = []

        def eval_spy(*a, **k):
            r = orig_eval(*a, **k)
            evals.append([float(v) for v in r])
            return r

        # The shipped end2end_eval cannot finish unless flag == 'finetune': with any other flag entropy_per_codec
        # stays empty and `min_len * entropy_per_codec[::2]` (nsc_module:755) is a (10,)x(0,) broadcast error, and in
        # the follower phase `len(_interested_var[4])` (:716) is len() of a scalar.  Validation is out of scope
        # (SURVEY 2 #7); for those two phases the call is answered by a scripted tuple so that the reference's tau
        # rule (:494-517), journal line (:520-530) and checkpoint naming (:548) still execute.  The finetune phase
        # runs the reference's validation loop for real (batch-of-1 hard-mode frames, per-codec entropies).
        script = {"n": 0, "ent": [2.5, 2.0, 2.2, 2.9, 1.1]}

        def eval_scripted(*a, **k):
            e = script["ent"][script["n"] % len(script["ent"])]
            script["n"] += 1
            r = (10.0 + script["n"], 9.0, 0.5, 1.0, 0.9, 7.25, 0.0, 0.0, e)
            evals.append([float(v) for v in r])
            return r

        for phase, call in (("one_ae", lambda: obj.one_ae()),
                            ("follower", lambda: obj._greedy_followers(1)),
                            ("finetune", lambda: obj._finetuning(2))):
            rec = PhaseRecorder(E)
            del E.shim.AdamOptimizer.OPTIMIZERS[:]
            E.shim.TRAIN_HOOKS[:] = [rec.hook]
            del evals[:]
            obj.end2end_eval = eval_spy if phase == "finetune" else eval_scripted
            n_saved = len(E.shim.SAVE_LOG)
            with quiet() as buf:
                call()
            pre = f"td_{phase}_"
            out[pre + "x"] = np.stack([s["x"] for s in rec.steps], 0)
            out[pre + "loss"] = np.stack([np.ravel(s["loss"]) for s in rec.steps], 0)
            out[pre + "loss_shape"] = np.array(json.dumps([list(np.shape(s["loss"])) for s in rec.steps]))
            out[pre + "tau"] = np.stack([np.ravel(s["tau"]) for s in rec.steps], 0)
            out[pre + "qon"] = np.array([float(s["is_quan_on"]) for s in rec.steps])
            out[pre + "lr"] = np.array([float(s["lr"]) for s in rec.steps])
            out[pre + "opt"] = np.array([s["opt"] for s in rec.steps])
            out[pre + "nvars"] = np.array([s["n_vars"] for s in rec.steps])
            out[pre + "evals"] = np.array(evals)
            for oid, g in rec.first_grads.items():
                for k, v in g.items():
                    if v is not None:
                        out[f"{pre}grad{oid}|{k}"] = v
            saves = E.shim.SAVE_LOG[n_saved:]
            out[pre + "saved_as"] = np.array(json.dumps([p for p, _ in saves]))
            for k, v in snapshot(os.path.join(work, saves[-1][0].lstrip("./") + ".npz")).items():
                out[f"{pre}ckpt|{k}"] = v
            out[pre + "stdout_tau_lines"] = np.array(json.dumps(
                [ln for ln in buf.getvalue().splitlines() if ln.startswith("Tau:")]))
        # ---- mode '0': _feedforward(2) -> cmrl_eval: utterance -> frames -> 2-codec cascade (soft: the_share 1.0,
        # cmrl.py:592) -> Hann overlap-add (cmrl.py:566-597), restored from the finetune checkpoint
        tests = [f32(0.03 * np.random.default_rng(90 + i).standard_normal(n)) for i, n in enumerate((1500, 993, 513))]
        obj._sep_test = ["test_%d.wav" % i for i in range(len(tests))]
        obj._load_sig = lambda f: (tests[int(f.split("_")[1].split(".")[0])].copy(), 1.0)
        os.makedirs(os.path.join(work, "end2end_performance"))
        written = []
        E.R.sf = type("sf", (), {"write": staticmethod(lambda path, sig, *a, **k: written.append((path, np.array(sig))))})
        with quiet() as buf:
            obj._feedforward(2)
        for i, (path, sig) in enumerate(written):
            out[f"td_ff_test{i}_in"], out[f"td_ff_test{i}_out"] = tests[i], sig
        out["td_ff_paths"] = np.array(json.dumps([p for p, _ in written]))
        out["td_ff_stdout"] = np.array(json.dumps([ln for ln in buf.getvalue().splitlines() if ln.startswith("Test Utt")]))
        out["td_journal"] = np.array(open(os.path.join(work, "doc", "1234567end2endcascade_journal.txt")).read())
        out["td_files"] = np.array(json.dumps(sorted(os.listdir(work)) + sorted(os.listdir(os.path.join(work, "check")))))
    finally:
        os.chdir(cwd)
        E.shim.TRAIN_HOOKS[:] = []


def run_lpc_phases(E, out):
    """one_ae_lpc -> _greedy_followers_lpc(1) -> _finetuning_lpc(2) with the residual fed (nsc_module:586-595)."""
    tf, M, U = E.tf, E.M, E.U
    win = U.utterance_to_segment(np.ones(1200), False)[0]
    raw = synth_frames(6, 71, win)
    res = synth_frames(6, 72, win)
    lsf = f32(np.sort(np.random.default_rng(73).uniform(0.03, 3.1, (6, 16)), axis=1))
    tr = np.concatenate([raw, lsf, res], 1)
    out["lp_tr_data"] = tr
    work = tempfile.mkdtemp(prefix="nsc_ref_lp_")
    os.makedirs(os.path.join(work, "check"))
    os.makedirs(os.path.join(work, "doc"))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        random.seed(6)
        np.random.seed(6)
        obj = new_cmrl(E, _the_strides=[2, 2], _epoch_tanh=2, _res_scalar=2.0, _rand_model_id="7654321",
                       _base_model_id="7654321", _target_entropy=2.2)
        obj._tr_data = tr.copy()
        # end2end_eval_lpc needs audiolazy.lpc / spectrum.poly2lsf (LPC *analysis* of validation wavs: out of scope);
        # it is answered by a scripted tuple so that the reference's LPC tau rule (:630-639) still executes.
        script = {"n": 0, "ent": [2.5, 2.0, 2.23, 2.26, 1.0]}
        evals = []

        def eval_scripted(*a, **k):
            e = script["ent"][script["n"] % len(script["ent"])]
            script["n"] += 1
            r = (10.0 + script["n"], 0.5, 1.0, 0.9, 7.25, 0.0, 0.0, e)
            evals.append([float(v) for v in r])
            return r

        obj.end2end_eval_lpc = eval_scripted
        for phase, call in (("one_ae", lambda: obj.one_ae_lpc()),
                            ("follower", lambda: obj._greedy_followers_lpc(1)),
                            ("finetune", lambda: obj._finetuning_lpc(2))):
            rec = PhaseRecorder(E)
            del E.shim.AdamOptimizer.OPTIMIZERS[:]
            E.shim.TRAIN_HOOKS[:] = [rec.hook]
            del evals[:]
            n_saved = len(E.shim.SAVE_LOG)
            with quiet() as buf:
                call()
            pre = f"lp_{phase}_"
            for key in ("x", "lpc_x", "res_x"):
                out[pre + key] = np.stack([s[key] for s in rec.steps], 0)
            out[pre + "loss"] = np.stack([np.ravel(s["loss"]) for s in rec.steps], 0)
            out[pre + "tau"] = np.array([float(np.ravel(s["tau"])[0]) for s in rec.steps])
            out[pre + "qon"] = np.array([float(s["is_quan_on"]) for s in rec.steps])
            out[pre + "lr"] = np.array([float(s["lr"]) for s in rec.steps])
            out[pre + "opt"] = np.array([s["opt"] for s in rec.steps])
            out[pre + "nvars"] = np.array([s["n_vars"] for s in rec.steps])
            out[pre + "evals"] = np.array(evals)
            for oid, g in rec.first_grads.items():
                for k, v in g.items():
                    if v is not None:
                        out[f"{pre}grad{oid}|{k}"] = v
            saves = E.shim.SAVE_LOG[n_saved:]
            out[pre + "saved_as"] = np.array(json.dumps([p for p, _ in saves]))
            for k, v in snapshot(os.path.join(work, saves[-1][0].lstrip("./") + ".npz")).items():
                out[f"{pre}ckpt|{k}"] = v
            out[pre + "stdout_tau_lines"] = np.array(json.dumps(
                [ln for ln in buf.getvalue().splitlines() if ln.startswith("tau:")]))
        out["lp_journal"] = np.array(open(os.path.join(work, "doc", "7654321end2endcascade_journal.txt")).read())
    finally:
        os.chdir(cwd)
        E.shim.TRAIN_HOOKS[:] = []


def run_lpc_utils(E, out):
    """lpc_utilities.py:28-77, 137-156 run on seeded inputs (with the restated ZFilter / lsf2poly of ref_env.py)."""
    P = E.P
    rng = np.random.default_rng(81)
    table = np.sort(np.asarray(E.C.lpc_coeff_lsf_bins, np.float64))
    lsf = np.stack([np.sort(rng.choice(table, 16, replace=False)) for _ in range(3)], 0).astype(np.float32)
    poly = P.lsf2poly_after_quan(lsf, 16)
    win = E.U.utterance_to_segment(np.ones(1200), False)[0]
    x = synth_frames(3, 82, win).astype(np.float32)[..., None]
    res = P.lpc_analysis_get_residual(x, poly)
    syn = P.lpc_synthesizer_tr(poly, res)
    out["lu_lsf"], out["lu_poly"], out["lu_x"], out["lu_res"], out["lu_syn"] = lsf, poly, x, res, syn
    out["lu_poly_dtype"] = np.array(str(poly.dtype) + " " + str(res.dtype) + " " + str(syn.dtype))


def main():
    E = ref_env.install()
    out = {}
    run_ops(E, out)
    run_codec_graphs(E, out)
    run_feedforward4(E, out)
    run_time_domain_phases(E, out)
    run_lpc_phases(E, out)
    run_lpc_utils(E, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
