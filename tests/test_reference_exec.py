"""CPU: the oracle against outputs of the REFERENCE'S OWN CODE (tests/golden/reference_exec.npz, produced by
tests/golden/make_reference_exec.py running /root/reference on the lazy-graph tf stand-in).  This is what pins the
oracle: every assertion below compares an oracle value with a number the reference's Python computed.
Tolerance 1e-9 relative (both sides are float64; the only differences are summation orders)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O
from oracle import nsc_oracle_torch as OT

from _replay import BKD, FX, PhaseReplay, lsf_table, named_store

TOL = 1e-9


def close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.max(np.abs(a - b) / (np.abs(b) + 1e-6)) if a.size else 0.0
    assert err <= tol, f"{what}: max elementwise rel err {err:.3e}"


# ------------------------------------------------------------------ op level
@pytest.mark.parametrize("tag,key,nb", [("c", "q_code", 32), ("l", "q_lsf", 256)])
@pytest.mark.parametrize("alpha", [-300.0, -20.0])
def test_quantizer_matches_reference(tag, key, nb, alpha):
    """nn_core_operator.py:140-164 incl. the hard/soft switch, the is_quan_on blend and exact ties."""
    x = FX[key]
    bins = np.linspace(-1, 1, 32).astype(np.float32).astype(np.float64) if tag == "c" else lsf_table()
    for sh in (True, False):
        for qv in (1.0, 0.0, 0.5):
            p, out = O.scalar_softmax_quantization(x, alpha, bins, qv, sh)
            k = f"q_{tag}_a{int(-alpha)}_{'soft' if sh else 'hard'}_q{int(qv * 10)}"
            close(out, FX[k + "_out"], what=k)
            if sh and qv == 1.0:
                close(p, FX[k + "_p"], what=k + "_p")
                close(O.quan_loss(p), FX[k + "_quan"], what=k + "_quan")
                close(O.entropy_coding_loss(p), FX[k + "_ent"], what=k + "_ent")


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_gated_bottleneck_matches_reference(tag):
    """nn_core_operator.py:82-112 (the Cin = 1 case 'd' broadcasts the residual add)."""
    cin, wide, dil, flat = [int(v) for v in FX[f"gb_{tag}_cfg"]]
    x = FX[f"gb_{tag}_x"]
    ps = O.ParamStore(name_seeded=True)
    O.gated_bottleneck(x, ps, "gb_" + tag, wide, 20, 9, dil, bool(flat))
    for k in ps.params:
        if k.endswith("/bias"):
            ps.params[k] = O.name_seeded_uniform(k, ps.params[k].shape, 0.05)
    ps.begin_replay()
    y = O.gated_bottleneck(x, ps, "gb_" + tag, wide, 20, 9, dil, bool(flat))
    close(y, FX[f"gb_{tag}_y"], what="gated_bottleneck " + tag)


def test_losses_match_reference():
    """loss_terms_and_measures.py:77-84, 130-183, 257-267 (incl. an identical pair, where the 1e-7 floors decide)."""
    d, o, p = FX["ls_dec"], FX["ls_ori"], FX["ls_p"]
    close(O.mse_loss(d, o), FX["ls_mse"], what="mse_loss")
    close(O.mfcc_loss(d, o), FX["ls_mfcc"], what="mfcc_loss")
    close(O.quan_loss(p), FX["ls_quan"], what="quan_loss")
    close(O.entropy_coding_loss(p), FX["ls_ent"], what="entropy_coding_loss")
    close(O.tf_stft(d)[1].reshape(FX["ls_mag"].shape), FX["ls_mag"], what="tf_stft magnitude")


@pytest.mark.parametrize("key,strides", [("s2", [2]), ("s22", [2, 2])])
@pytest.mark.parametrize("kind", ["td", "lpc"])
def test_codec_graph_matches_reference(key, strides, kind):
    """computational_graph_end2end_quan_on / _lpc (nsc_module:262-335): soft, hard and unquantised decodes."""
    x = FX["cg_x"]
    ps = named_store(1, [strides], [32])
    names, shapes = json.loads(str(FX[f"cg_{key}_varnames"]))
    mine = [k for k in ps.params if k.startswith("scope_1/")]
    assert mine == names and [list(np.shape(ps.params[k])) for k in mine] == shapes   # creation order + shapes
    for sh in (True, False):
        ps.begin_replay()
        o = O.codec_forward(x, ps, "scope_1", BKD, strides, 32, 1.0, sh)
        k = f"cg_{key}_{kind}_{'soft' if sh else 'hard'}"
        close(o["decoded"], FX[k + "_dec"], what=k)
        close(o["code"][0, :, 0], FX[k + "_code0"], what=k + " code")
        if sh:
            close(o["p"], FX[k + "_p"], what=k + " p")
    ps.begin_replay()
    o = O.codec_forward(x, ps, "scope_1", BKD, strides, 32, 0.0, True)
    close(o["decoded"], FX[f"cg_{key}_{kind}_noquan_dec"], what="no-quan decode")


@pytest.mark.parametrize("tag,n,strides", [("ff4", 4, [2, 2]), ("ff2", 2, [2])])
def test_cascade_feedforward_matches_reference(tag, n, strides):
    """all_modules_feedforward (cmrl.py:513-543): config 4 (4 codecs x [2,2]) and config 5 (2 codecs x [2])."""
    x = FX[tag + "_x"]
    ps = named_store(n, [strides] * n, [32] * n)
    for sh in (True, False):
        ps.begin_replay()
        outs, dec = O.cascade_forward(x, ps, BKD, [strides] * n, [32] * n, 1.0, sh, res_scalar=2.0)
        want = FX[f"{tag}_{'soft' if sh else 'hard'}_yhat"]
        close(dec, want.sum(0), what=tag + " sum")
        ents = [O.entropy_coding_loss(o["p"]) for o in outs]
        close(ents, FX[f"{tag}_{'soft' if sh else 'hard'}_ent"], what=tag + " entropies")


# ------------------------------------------------------------------ phases (loss vectors, gradients, Adam, checkpoints)
def test_time_domain_phases_match_reference():
    """one_ae -> _greedy_followers(1) -> _finetuning(2): every step's [B] loss vector, the first-step gradient of each
    optimizer, and the checkpoint after each phase (nsc_module:891-939, 424-460; cmrl.py:22-135, 295-390)."""
    rp = PhaseReplay("td", lpc=False)
    rp.run("one_ae", num=1, mode="quan_last", train=["scope_1"])
    rp.reinit(["scope_2"])
    rp.run("follower", num=2, mode="quan_last", train=["scope_2"])
    rp.run("finetune", num=2, mode="finetune", train=["scope_1", "scope_2"])
    assert rp.checked["loss"] == 14 and rp.checked["grad_sets"] == 4 and rp.checked["ckpt"] == 3


def test_lpc_phases_match_reference():
    """one_ae_lpc -> _greedy_followers_lpc(1) -> _finetuning_lpc(2) with the residual fed (nsc_module:989-1073,
    586-595; cmrl.py:137-293, 392-511): 16/272 : 256/272 blend incl. the LSF entropy, LSF quantizer trained."""
    rp = PhaseReplay("lp", lpc=True, res_scalar=2.0)
    rp.run("one_ae", num=1, mode="one_ae_lpc", train=["lpc_quan", "scope_1"])
    rp.reinit(["scope_2"])
    rp.run("follower", num=2, mode="quan_last", train=["scope_2"])
    rp.run("finetune", num=2, mode="finetune_lpc", train=["lpc_quan", "scope_1", "scope_2"])
    assert rp.checked["loss"] == 12 and rp.checked["grad_sets"] == 4 and rp.checked["ckpt"] == 3


# ------------------------------------------------------------------ tau controllers, journal
def _tau_lines(pre):
    return json.loads(str(FX[pre + "stdout_tau_lines"]))


def test_tau_controller_time_domain_matches_reference():
    """nsc_module:494-517: +-0.015 toward target_entropy; finetune: tau_1 / tau_2 toward 1.5 / 2.5."""
    for ph in ("one_ae", "follower", "finetune"):
        ev = FX[f"td_{ph}_evals"]
        tau, t12 = 0.3, [0.3, 0.3]
        for e, line in zip(ev, _tau_lines(f"td_{ph}_")):
            tau, t12 = O.tau_update(ph if ph == "finetune" else "pretrain", tau, t12, e[8], e[6:8], 2.2)
            assert line == 'Tau: %7.5f, Tau_1: %7.5f, Tau_2: %7.5f' % (tau, t12[0], t12[1])
    # the taus fed to the NEXT epoch's steps are the controller's outputs (finetune feeds tau_1, tau_2)
    assert np.allclose(FX["td_finetune_tau"][2], [0.285, 0.285])


def test_tau_controller_lpc_matches_reference():
    """nsc_module:630-639: only while is_quan_on == 1; +0.015 above target + 0.05, -0.045 below target."""
    for ph, qons in (("one_ae", [0.0, 1.0]), ("follower", [1.0, 1.0]), ("finetune", [1.0, 1.0])):
        ev = FX[f"lp_{ph}_evals"]
        tau = 0.3
        got = []
        for e, q in zip(ev, qons):
            tau, _ = O.tau_update("x", tau, [0, 0], e[7], None, 2.2, lpc=True, is_quan_on=q)
            if q == 1.0:
                got.append("tau: " + str(tau))
        assert got == _tau_lines(f"lp_{ph}_")
    assert np.allclose(FX["lp_finetune_tau"], [0.3, 0.3, 0.255, 0.255])


def test_validation_entropy_matches_reference_finetune():
    """end2end_eval (nsc_module:657-758) in the finetune phase: frames fed one at a time, hard assignment, the entropy
    of each frame's own histogram, averaged per utterance then weighted by length.  Replayed from the finetune
    checkpoint state at the time of each validation (epoch 0 -> after 2 steps)."""
    rp = PhaseReplay("td", lpc=False)
    rp.run("one_ae", num=1, mode="quan_last", train=["scope_1"], check=False)
    rp.reinit(["scope_2"])
    rp.run("follower", num=2, mode="quan_last", train=["scope_2"], check=False)
    utts = FX["td_val_utts"]

    def validate():
        per_codec = []
        for u in utts:
            fr = O.utterance_to_segment(u, True)[..., None]
            ps = rp.as_store()
            outs, _ = O.cascade_forward(fr, ps, BKD, [[2], [2]], [32, 32], 1.0, False, res_scalar=1.0)
            per_codec.append([np.mean([O.entropy_coding_loss(o["p"][j:j + 1]) for j in range(len(fr))]) for o in outs])
        per_codec = np.array(per_codec)           # equal utterance lengths -> plain mean
        return per_codec.mean(0)

    seen = []
    rp.run("finetune", num=2, mode="finetune", train=["scope_1", "scope_2"], check=False,
           after_step=lambda s: seen.append(validate()) if s in (1, 3) else None)
    ev = FX["td_finetune_evals"]
    for got, want in zip(seen, ev):
        close(got, want[6:8], tol=1e-7, what="ent_codec_1/2")
        close(got.sum(), want[8], tol=1e-7, what="fully_entropy")


def test_utterance_inference_matches_reference():
    """mode '0' (_feedforward -> cmrl_eval, cmrl.py:566-597): utterance -> hop-480 frames -> 2-codec cascade with the SOFT
    assignment (the_share fed 1.0, cmrl.py:592) -> three-window Hann overlap-add; 3-, 2- and 1-frame utterances."""
    rp = PhaseReplay("td", lpc=False)
    rp.run("one_ae", num=1, mode="quan_last", train=["scope_1"], check=False)
    rp.reinit(["scope_2"])
    rp.run("follower", num=2, mode="quan_last", train=["scope_2"], check=False)
    rp.run("finetune", num=2, mode="finetune", train=["scope_1", "scope_2"], check=False)
    for i in range(3):
        u = FX[f"td_ff_test{i}_in"]
        fr = O.utterance_to_segment(u, True)[..., None]
        _, dec = O.cascade_forward(fr, rp.as_store(), BKD, [[2], [2]], [32, 32], 1.0, True, res_scalar=1.0)
        close(O.overlap_add(dec), FX[f"td_ff_test{i}_out"], tol=1e-7, what=f"utterance {i}")


def test_journal_and_checkpoint_names():
    """Formats for SURVEY 8f N4: journal lines (nsc_module:520-530, 623-628; cmrl.py:623-625) and checkpoint paths."""
    j = str(FX["td_journal"])
    assert 'Epoch   0: SNR: 11.00000 dB Si-SNR: 9.00000 dB STOI: 0.50000 PESQ: 1.00000 _quan_loss: 7.25000tau: 0.31500' \
           '   fully_entropy: 2.50000 \n' in j
    assert json.loads(str(FX["td_one_ae_saved_as"])) == ["./check/model_bnn_ac_1234567_.ckpt"]
    assert json.loads(str(FX["td_follower_saved_as"])) == ["./check/model_bnn_ac_1234567_follower_1end2endcascade.ckpt"]
    assert json.loads(str(FX["td_finetune_saved_as"])) == ["./check/model_bnn_ac_1234567_finetune_2end2endcascade.ckpt"]
    assert 'fully_snr: 0.00000   fully_pesq: 0.00000  fully_entropy: 2.50000 \n' in str(FX["lp_journal"])


# ------------------------------------------------------------------ LPC utilities (N3)
def test_lpc_utilities_match_reference():
    """lpc_utilities.py:28-33 (lsf2poly_after_quan), :37-77 (7 cross-faded sub-frame FIR residual), :137-156 (IIR synthesis)."""
    poly = O.lsf2poly_after_quan(FX["lu_lsf"])
    assert poly.dtype == np.float32 and np.array_equal(poly, FX["lu_poly"])
    res = O.lpc_analysis_get_residual(FX["lu_x"], FX["lu_poly"])
    close(res, FX["lu_res"], tol=1e-5, what="residual")       # both sides round to float32 at the end
    syn = O.lpc_synthesizer_tr(FX["lu_poly"], FX["lu_res"])
    close(syn, FX["lu_syn"], tol=1e-5, what="synthesis")
