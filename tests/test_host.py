"""CPU tests: the C-ABI library loads and exports every symbol include/nsc_hip.h declares (no compute calls),
host logic (CLI, layouts, framing, scopes) and the data-parallel path over gloo with world_size 2."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nsc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


def test_library_exports_every_declared_symbol():
    from nsc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nsc_hip.h")).read()
    declared = set(re.findall(r"\b(nsc_[a-z0-9_]+)\s*\(", hdr)) - {"nsc_conv_desc"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nsc_version() >= 100
    nm = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, nm), f"{name} not exported"


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nsc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, "main.py")).read(), re.M)


def test_compute_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nsc_amd import _lib
    from nsc_amd.engine import CascadeEngine
    with pytest.raises(_lib.NscError):
        CascadeEngine(2, 1)
    from nsc_amd import nn_core_operator as nn
    with pytest.raises(_lib.NscError):
        nn.activation_func(torch.zeros(1, 4, 1))


def test_cli_has_the_reference_flags():
    import main
    opts = {a.option_strings[0] for a in main.build_parser()._actions if a.option_strings}
    ref = {"--learning_rate_tanh", "--learning_rate_greedy_followers", "--epoch_tanh", "--epoch_greedy_followers",
           "--from_where_step", "--batch_size", "--num_resnets", "--training_mode", "--base_model_id", "--suffix",
           "--window_size", "--bottleneck_kernel_and_dilation", "--is_cq", "--the_strides", "--save_unique_mark",
           "--coeff_term", "--res_scalar", "--pretrain_step", "--target_entropy", "--num_bins_for_follower"}
    assert ref <= opts and len(ref) == 20   # main.py:6-27 (20 add_argument calls + is_cq = the "21 flags" incl. help)


def test_op_surface_names_and_signatures_match_reference():
    import inspect
    from nsc_amd import nn_core_operator as nn, loss_terms_and_measures as L
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(nn.conv1d) == ["inputs", "num_filters", "filter_size", "padding", "dilation_rate", "strides", "activation"]
    assert sig(nn.conv1d_depth) == sig(nn.conv1d)
    assert sig(nn.change_channel) == ["the_input", "wide_layer", "the_channel", "kernel_size", "dilation_rate", "strides", "activation"]
    assert sig(nn.gated_bottleneck) == ["the_input", "wide_layer", "narrow_layer", "non_dilated_neck_kernel_size",
                                        "dilated_neck_kernel_size", "dilation_rate", "is_last_flat", "the_share"]
    assert sig(nn.scalar_softmax_quantization) == ["floating_code", "alpha", "bins", "is_quan_on", "the_share", "code_length", "num_kmean_kernels"]
    for n in ("activation_func", "batch_norm", "the_bottleneck", "gated_bottleneck_decoder", "vector_softmax_quantization"):
        assert hasattr(nn, n)
    for n in ("mse_loss", "mse_loss_v1", "mfcc_loss", "tf_stft", "quan_loss", "entropy_coding_loss", "entropy_to_bitrate",
              "bitrate_to_entropy", "snr", "si_snr"):
        assert hasattr(L, n)
    d = inspect.signature(nn.gated_bottleneck).parameters
    assert (d["wide_layer"].default, d["narrow_layer"].default, d["dilation_rate"].default) == (30, 10, 1)


def test_host_helpers_match_reference_kats():
    from nsc_amd import utilities as U, loss_terms_and_measures as L, constants as K
    utt = np.random.default_rng(KATS["utt_seed"]).standard_normal(KATS["utt_len"])
    assert np.array_equal(U.utterance_to_segment(utt, True), np.array(KATS["seg_post"]))
    assert np.array_equal(U.utterance_to_segment(utt, False), np.array(KATS["seg_win"]))
    for n, cnt in KATS["frame_counts"].items():
        assert U.num_frames(int(n)) == cnt
    ones = np.ones(512)
    for i, key in enumerate(("hann_first", "hann_mid", "hann_last")):
        assert np.array_equal(U.hann_process(ones, i, 3), np.array(KATS[key]))
        assert np.array_equal(U.hann_windows3()[i], np.array(KATS[key], np.float32))
    for e, s, v in KATS["entropy_to_bitrate"]:
        assert L.entropy_to_bitrate(e, s) == v
    for b, s, v in KATS["bitrate_to_entropy"]:
        assert L.bitrate_to_entropy(b, s) == v
    assert K.lpc_coeff_lsf_bins == KATS["lsf_bins"]
    assert np.array_equal(L.mel_matrix_cat(), O.mel_matrix_cat().astype(np.float32))


@pytest.mark.parametrize("key,strides", [("2", [2]), ("2_2", [2, 2])])
def test_engine_layout_matches_reference_topology(key, strides):
    from nsc_amd.engine import CascadeEngine
    eng = CascadeEngine(4, 1, strides=[strides], layout_only=True)
    topo = KATS["topology"][key]
    ref = []
    for op in topo["encoder"] + topo["decoder"]:
        if op[0] == "conv1d":
            ref.append((op[3], op[1][2], op[2]))
        elif op[0] == "separable_conv1d":
            ref.append(("dw", op[3], op[1][2]))
            ref.append((1, op[1][2], op[2]))
    mine = []
    for name, (off, shape) in eng.layout.entries.items():
        if name.endswith("depthwise_kernel"):
            mine.append(("dw", shape[0], shape[1]))
        elif name.endswith("kernel"):
            mine.append(tuple(shape))
    assert mine == ref
    assert eng.layout.size == {"2": 350185, "2_2": 540400}[key]
    assert eng.codecs[0].L == topo["code_shape"][1]


def test_variable_store_creation_order_and_replay():
    from nsc_amd.scope import VariableStore, variable_scope, set_store
    st = VariableStore(device="cpu")
    set_store(st)
    with variable_scope("scope_1"):
        n1, n2 = st.uniq("conv1d"), st.uniq("conv1d")
        st.get(n1 + "/kernel", (3, 2, 4), st.glorot(6, 12))
        st.get(n2 + "/kernel", (1, 4, 4), st.glorot(4, 4))
    assert list(st.vars) == ["scope_1/conv1d/kernel", "scope_1/conv1d_1/kernel"]
    st.begin_pass()
    with variable_scope("scope_1"):
        assert st.uniq("conv1d") == "scope_1/conv1d"
        with pytest.raises(ValueError):
            st.get("scope_1/conv1d/kernel", (9, 9, 9), st.glorot(1, 1))
    assert len(st.trainable_variables("scope_1")) == 2 and st.trainable_variables("scope_2") == []
    set_store(None)


# --------------------------------------------------------------------------------------------------
# data-parallel path, world_size 2 over gloo (CPU)
# --------------------------------------------------------------------------------------------------
_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
from nsc_amd.dist import Comm
from oracle import nsc_oracle as O, nsc_oracle_torch as OT
from tests._util import BKD, make_store, synth_frames
comm = Comm(backend="gloo")
assert comm.world == 2
B = 4
ps = make_store(1, [[2]], [32])
x = synth_frames(B)
lo, hi = comm.shard(B)
assert (lo, hi) == (comm.rank * 2, comm.rank * 2 + 2)
coeff, tau = [60.0, 10.0, 10.0, 0.0], 0.4
# local forward on this rank's frames; GLOBAL entropy through a histogram all-reduce (SURVEY 8e (2))
tp = OT.TorchParams(ps)
xt = torch.tensor(x[lo:hi])
outs, dec = OT.cascade_forward(xt, tp, BKD, [[2]], 1.0, True)
p = outs[0]["p"]
hist = p.detach().reshape(-1, 32).sum(0)
ghist = hist.clone()
comm.allreduce(ghist)
extra = ghist - hist                      # other ranks' histogram enters as a constant
loss = (coeff[0] * OT.mse_loss(dec, xt[:, :, 0]) + coeff[1] * OT.mfcc_loss(dec, xt[:, :, 0]) + coeff[2] * OT.quan_loss(p)).sum() \
    + B * tau * OT.entropy_coding_loss(p, hist_extra=extra)
loss.backward()
flat = torch.cat([tp.t[k].grad.reshape(-1) for k in tp.names])
comm.allreduce(flat)                      # SUM over ranks (the reference's vector loss sums over the batch)
if comm.rank == 0:
    tp2 = OT.TorchParams(ps)
    o2, d2 = OT.cascade_forward(torch.tensor(x), tp2, BKD, [[2]], 1.0, True)
    OT.total_loss_sum(d2, torch.tensor(x)[:, :, 0], [o2[0]["p"]], coeff, tau, "quan_last").backward()
    ref = torch.cat([tp2.t[k].grad.reshape(-1) for k in tp2.names])
    err = float((flat - ref).abs().max() / ref.abs().max())
    print(json.dumps({"err": err, "max": float(comm.max_float(3.0 + comm.rank, "cpu"))}))
else:
    comm.max_float(3.0 + comm.rank, "cpu")
comm.barrier()
comm.close()
'''


def test_data_parallel_two_ranks_gloo_equals_single_process(tmp_path):
    """2 ranks x 2 frames with summed gradients + all-reduced histogram == 1 process x 4 frames (float64 oracle)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["err"] < 1e-9, res
    assert res["max"] == 4.0


# ---------------------------------------------------------------------------------------------------------------------
# host logic of the trainer against what the reference's own code printed / wrote (tests/golden/reference_exec.npz)
# ---------------------------------------------------------------------------------------------------------------------
def _fx():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_exec.npz"))


def test_tau_controller_values_match_reference_stdout():
    """nsc_amd tau_controller == the 'Tau:' / 'tau:' lines the reference's model_training[_lpc] printed for scripted and
    real validation entropies: +-0.015 toward target (2.5 -> up, 2.0 -> down, 2.2 -> hold), finetune tau_1/tau_2 toward
    1.5 / 2.5, LPC +0.015 above target+0.05 and -0.045 below target, nothing while is_quan_on == 0."""
    from nsc_amd.neural_speech_coding_module import tau_controller
    fx = _fx()
    for ph in ("one_ae", "follower", "finetune"):
        taus = (0.3, 0.3, 0.3)
        lines = json.loads(str(fx[f"td_{ph}_stdout_tau_lines"]))
        assert len(lines) == len(fx[f"td_{ph}_evals"]) > 0
        for e, line in zip(fx[f"td_{ph}_evals"], lines):
            taus = tau_controller('finetune' if ph == "finetune" else 'pretrain', False, 1.0, taus, e[8], list(e[6:8]), 2.2)
            assert line == 'Tau: %7.5f, Tau_1: %7.5f, Tau_2: %7.5f' % taus
    for ph, qons in (("one_ae", [0.0, 1.0]), ("follower", [1.0, 1.0]), ("finetune", [1.0, 1.0])):
        tau, got = 0.3, []
        for e, q in zip(fx[f"lp_{ph}_evals"], qons):
            tau = tau_controller('x', True, q, (tau, 0.0, 0.0), e[7], None, 2.2)[0]
            if q == 1.0:
                got.append("tau: " + str(tau))
        assert got == json.loads(str(fx[f"lp_{ph}_stdout_tau_lines"]))


def test_journal_lines_match_reference_format():
    """Given the reference's numbers, journal_line / journal_line_lpc reproduce the reference's journal text byte for byte."""
    from nsc_amd.neural_speech_coding_module import journal_line, journal_line_lpc
    fx = _fx()
    j = str(fx["td_journal"])
    e = fx["td_one_ae_evals"][0]          # (snr, si_snr, stoi, pesq, linearity, quan, ent1, ent2, fully_entropy)
    assert journal_line(0, e[0], e[1], e[2], e[3], e[5], 0.315, e[8]) in j
    e = fx["td_finetune_evals"][1]
    assert journal_line(1, e[0], e[1], e[2], e[3], e[5], 0.3, e[8]) in j
    jl = str(fx["lp_journal"])
    e = fx["lp_one_ae_evals"][0]          # (snr, stoi, pesq, linearity, quan, fully_snr, fully_pesq, fully_entropy)
    assert journal_line_lpc(0, e[0], e[1], e[2], e[4], e[5], e[6], e[7]) in jl


def test_loss_configs_of_the_cli_follow_the_reference():
    """_loss_cfgs: finetune quan weight is coeff[2] x global batch (cmrl.py:355 sums over the batch before broadcasting);
    one_ae_lpc always blends 16 : L incl. the tau slot (nsc_module:1032-1050); finetune_lpc always trains the LSF
    quantizer (cmrl.py:398-401, 464-466) - and it is the step bench.py times."""
    from nsc_amd.neural_speech_coding_module import neuralSpeechCodingModule
    import bench

    def mod(lpc, is_cq, B=128):
        m = neuralSpeechCodingModule.__new__(neuralSpeechCodingModule)
        m._coeff_term, m._is_pure_time_domain, m._is_cq, m._the_strides = [60.0, 10.0, 10.0, 0.0], not lpc, is_cq, [2, 2]
        m._batch_size, m._comm = B, None
        return m

    _, quan, tau_map = mod(False, 0)._loss_cfgs(2, "finetune")
    assert quan["c_quan"] == [10.0 * 128, 10.0 * 128] and [t[3] for t in tau_map] == [1, 2]
    for is_cq in (0, 1):
        _, quan, tau_map = mod(True, is_cq)._loss_cfgs(1, "single")
        assert abs(quan["c_quan"][0] - 10.0 * 256 / 272) < 1e-12
        assert ("c_ent", 0, 256.0 / 272.0, 0) in tau_map
        assert (quan.get("c_quan_lpc", 0.0) != 0.0) == bool(is_cq) and bool(quan.get("train_lpc")) == bool(is_cq)
        _, quan, tau_map = mod(True, is_cq)._loss_cfgs(2, "finetune")
        assert quan["c_quan"] == [10.0, 10.0] and quan["c_quan_lpc"] == 10.0 and quan["train_lpc"] and tau_map == []
        ref = bench.step_cfg()
        for k in ("is_quan_on", "c_time", "c_freq", "c_quan", "c_ent", "trainable", "slot", "c_quan_lpc", "train_lpc", "quan_op",
                  "global_entropy"):
            assert quan[k] == ref[k], k


def test_mel_band_ranges_cover_every_nonzero():
    """nsc_recon_loss_banded visits only [k_lo, k_hi) of a mel column and [j_lo, j_hi) of a bin within each bank: every
    non-zero of the matrix must lie inside, and the ranges must stay inside their bank."""
    import numpy as np
    from nsc_amd.loss_terms_and_measures import MEL_BANKS, mel_band_ranges, mel_matrix_cat
    m = mel_matrix_cat()
    r = mel_band_ranges(m)
    cols, rows = r[:2 * m.shape[1]].reshape(-1, 2), r[2 * m.shape[1]:].reshape(m.shape[0], len(MEL_BANKS), 2)
    inside_c = np.zeros_like(m, dtype=bool)
    inside_r = np.zeros_like(m, dtype=bool)
    edges = np.concatenate([[0], np.cumsum(MEL_BANKS)])
    for j, (lo, hi) in enumerate(cols):
        inside_c[lo:hi, j] = True
    for k in range(m.shape[0]):
        for b in range(len(MEL_BANKS)):
            lo, hi = rows[k, b]
            assert lo == hi or (edges[b] <= lo and hi <= edges[b + 1])
            inside_r[k, lo:hi] = True
    nz = m != 0
    assert not (nz & ~inside_c).any() and not (nz & ~inside_r).any()
    assert inside_c.sum() < 0.06 * m.size and inside_r.sum() < 0.06 * m.size      # ~1.9 k of 47 k terms


def test_epoch_rows_shard_like_a_single_process_at_the_global_batch():
    """Data parallel feed of the trainer (nsc_amd/neural_speech_coding_module.py: epoch_permutation / epoch_batches): in
    every step the ranks' rows are disjoint and their concatenation (rank order) is exactly the step of a 1-process run at
    batch world * B; the order is a function of (seed, epoch) only, and one epoch's shuffle acts on the previous order like
    the reference's in-place np.random.shuffle of the matrix (nsc_module:115-121, 460)."""
    from nsc_amd.neural_speech_coding_module import epoch_batches, epoch_permutation
    n, B, world, seed = 203, 4, 2, 7
    perm = None
    for epoch in range(3):
        perm = epoch_permutation(n, seed, epoch, perm)
        assert sorted(perm.tolist()) == list(range(n))
        one = epoch_batches(perm, B * world, 1, 0, seed, epoch, 2500)
        r0 = epoch_batches(perm, B, world, 0, seed, epoch, 2500)
        r1 = epoch_batches(perm, B, world, 1, seed, epoch, 2500)
        assert len(one) == len(r0) == len(r1) == len(range(0, n - B * world, B * world))
        for a, b, c in zip(one, r0, r1):
            assert np.array_equal(a, np.concatenate([b, c])) and not set(b.tolist()) & set(c.tolist())
        # contiguous rows of the (shuffled) matrix, shuffled batch starts: every step is perm[i : i + 2B] for a distinct i
        starts = sorted(int(np.where(perm == a[0])[0][0]) for a in one)
        assert starts == list(range(0, n - B * world, B * world))
        again = epoch_batches(epoch_permutation(n, seed, 0), B, world, 1, seed, 0, 3)
        assert len(again) == 3
    p0 = epoch_permutation(n, seed, 0)
    assert np.array_equal(p0, np.arange(n))                       # epoch 0 reads the matrix in file order like the reference
    assert not np.array_equal(epoch_permutation(n, seed, 1, p0.copy()), np.arange(n))
    assert np.array_equal(epoch_permutation(n, seed, 1, np.arange(n)), epoch_permutation(n, seed, 1, np.arange(n)))


def test_bench_starts_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a launcher environment must start the N ranks itself.  On this CPU
    box: with the RCCL backend and no GPU the parent refuses with a clear message and rc 2 (nothing launched); with
    NSC_DIST_BACKEND=gloo it launches torch.distributed.run as a child, whose ranks fail for lack of a GPU - the parent relays
    the failure as a non-zero rc and prints no JSON line."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NSC_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and "{" not in r.stdout
    env["NSC_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert "launching 2 ranks" in r.stderr
    if not torch.cuda.is_available():
        assert r.returncode != 0 and '"metric"' not in r.stdout


def test_comm_refuses_rccl_with_fewer_gpus_than_local_ranks(monkeypatch):
    """First-contact robustness: RCCL with two ranks on one device fails deep inside the first collective; Comm says so up front."""
    from nsc_amd import dist as D
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setattr(D.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(D.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="one GPU per local rank"):
        D.Comm(backend="nccl")


def _bf16_split3(a):
    """x = hi + lo + lo2 with round-to-nearest-even bf16 pieces (csrc/nsc_common.h: nsc_split2), in numpy."""
    def rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
        return (u & 0xffffffff).astype(np.uint32).view(np.float32)
    hi = rne(a)
    lo = rne(a - hi)
    lo2 = rne(a - hi - lo)
    return hi, lo, lo2


@pytest.mark.parametrize("which", [0, 1])
def test_split_conv_image_index_rebuilds_the_gemm_operand_on_the_host(which):
    """nsc_conv1d_simage_index is host code: emulate nsc_gather's word rule (bits 26..29 = piece plane and source stride, csrc/misc.hip:
    gather_word) in numpy and check that the image IS the GEMM's A operand in fragment order - forward: Wmat[o][tap * 100 + ci] =
    W[tap][ci][o]; data gradient (polyphase): Wmat[2 ci + p][t' * 100 + o] = W[7 - 2 t' + p][ci][o], zero where that tap does not exist -
    as three bf16 pieces that sum back to the float32 weight to 2^-24."""
    import ctypes as C
    from nsc_amd import _lib
    from nsc_amd._lib import ConvDesc
    lib = _lib.load()
    d = ConvDesc(B=4, Cin=100, Cout=100, Tin=512, Tout=256, K=9, dil=1, stride=2, padL=3, act=0, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                 accumulate=0)
    n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
    assert n > 0 and n % 256 == 0
    off = 12
    idx = np.empty(n, np.int32)
    assert lib.nsc_conv1d_simage_index(which, C.byref(d), off, idx.ctypes.data_as(C.c_void_p)) == 0
    rng = np.random.default_rng(which)
    w = rng.standard_normal((9, 100, 100)).astype(np.float32)
    src = np.concatenate([np.zeros(off, np.float32), w.reshape(-1), np.zeros(128, np.float32)])
    pieces = _bf16_split3(src)
    strides = [20, 25, 50, 100, 1]
    M, KT, nrt = (100, 9, 7) if which == 0 else (200, 5, 13)
    ks = n // (nrt * 3 * 256)
    got = np.zeros((3, nrt * 16, ks * 32), np.float32)              # piece, row m, k
    im = idx.reshape(ks, nrt, 3, 64, 4)
    for s in range(ks):
        for rt in range(nrt):
            for p in range(3):
                e = im[s, rt, p]                                     # [lane][word]
                m_ = (e >> 26) & 15
                base = e & 0x3ffffff
                valid = e >= 0
                assert np.all(~valid | (m_ > 0))                    # every entry of these images is a packed pair of pieces
                plane = np.where(valid, (m_ - 1) // 5, 0)
                assert np.all(~valid | (plane == p))
                stride = np.array(strides)[np.where(valid, (m_ - 1) % 5, 0)]
                for lane in range(64):
                    row = rt * 16 + (lane & 15)
                    for jw in range(4):
                        k = 32 * s + 8 * (lane >> 4) + 2 * jw
                        if valid[lane, jw]:
                            got[p, row, k] = pieces[p][base[lane, jw]]
                            got[p, row, k + 1] = pieces[p][base[lane, jw] + stride[lane, jw]]
    want = np.zeros((nrt * 16, ks * 32), np.float32)
    for tp in range(KT):
        if which == 0:
            want[:100, tp * 100:(tp + 1) * 100] = w[tp].T             # [o][ci]
        else:
            for par in range(2):
                kk = 7 - 2 * tp + par
                if 0 <= kk < 9:
                    want[par:200:2, tp * 100:(tp + 1) * 100] = w[kk]  # [ci][o] -> rows 2 ci + par
    rebuilt = got.astype(np.float64).sum(0)
    assert np.max(np.abs(rebuilt - want)) <= 2.0 ** -23 * np.max(np.abs(want))
    assert np.array_equal(got[0], _bf16_split3(want)[0])              # the leading piece is the bf16 rounding of the weight itself


@pytest.mark.parametrize("C_", [100, 50])
def test_three_launch_dgrad_image_is_the_weight_pieces_in_kernel_layout(C_):
    """nsc_gated_block_simage_index(which = 2) is host code: with nsc_gather's word rule emulated in numpy the image must be, per half
    of 50 output channels, the three bf16 pieces of W9 as [tap][ci][56] (+ 8 zeros), then those of Wl | Wr as [tap][ci][40] (+ 8 zeros) -
    pieces that sum back to the float32 weight to 2^-24, pad columns and tails structural zeros (csrc/block_bwd_split.hip)."""
    import ctypes as C
    from nsc_amd import _lib
    lib = _lib.load()
    n = int(lib.nsc_gated_block_simage_words(2, C_, C_, 1))
    assert n > 0 and n == int(lib.nsc_gated_block_simage_words(2, C_, 1, 2))         # the image does not depend on Cin / dilation
    rng = np.random.default_rng(C_)
    shapes = [(1, C_, 20), (20,), (15, 20, 20), (20,), (15, 20, 20), (20,), (9, 20, C_), (C_,)]
    w = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
    offs = np.concatenate([[0], np.cumsum([a.size for a in w])[:-1]]).astype(np.int64) + 12
    src = np.concatenate([np.zeros(12, np.float32)] + [a.reshape(-1) for a in w] + [np.zeros(64, np.float32)])
    idx = np.empty(n, np.int32)
    assert lib.nsc_gated_block_simage_index(2, C_, C_, 1, (C.c_long * 8)(*[int(o) for o in offs]), idx.ctypes.data_as(C.c_void_p)) == 0
    pieces = _bf16_split3(src)
    valid = idx >= 0
    m_ = np.where(valid, (idx >> 26) & 15, 0)
    base = np.where(valid, idx & 0x3ffffff, 0)
    assert np.all(~valid | ((m_ - 1) % 5 == 4))                                     # every pair is (o, o + 1): gather stride 1
    plane = np.where(valid, (m_ - 1) // 5, 0)
    lo = np.where(valid, np.choose(plane, [p[base] for p in pieces]), 0.0)          # low half of the word = first element of the pair
    hi = np.where(valid, np.choose(plane, [p[base + 1] for p in pieces]), 0.0)
    el = np.stack([lo, hi], -1).reshape(-1)                                          # the image as bf16-valued elements
    plw9, plw15 = 9 * 20 * 56 + 8, 15 * 20 * 40 + 8
    nh = C_ // 50
    assert el.size == nh * 3 * plw9 + 3 * plw15
    pl_of = np.stack([np.where(valid, plane, -1)] * 2, -1).reshape(-1)
    for hf in range(nh):
        got = el[hf * 3 * plw9:(hf + 1) * 3 * plw9].reshape(3, plw9)
        pls = pl_of[hf * 3 * plw9:(hf + 1) * 3 * plw9].reshape(3, plw9)
        for p in range(3):
            assert np.all((pls[p] == p) | (pls[p] == -1))
        body = got[:, :9 * 20 * 56].reshape(3, 9, 20, 56)
        assert np.all(got[:, 9 * 20 * 56:] == 0) and np.all(body[..., 50:] == 0)
        want = w[6][:, :, 50 * hf:50 * hf + 50]
        assert np.max(np.abs(body[..., :50].astype(np.float64).sum(0) - want)) <= 2.0 ** -23 * np.max(np.abs(want))
        assert np.array_equal(body[0, ..., :50], _bf16_split3(want)[0])
    got = el[nh * 3 * plw9:].reshape(3, plw15)
    body = got[:, :15 * 20 * 40].reshape(3, 15, 20, 40)
    assert np.all(got[:, 15 * 20 * 40:] == 0)
    want = np.concatenate([w[2], w[4]], axis=2)                                      # [tap][ci][lin 0..19 | gate 20..39]
    assert np.max(np.abs(body.astype(np.float64).sum(0) - want)) <= 2.0 ** -23 * np.max(np.abs(want))



# ---- TF V2 tensor-bundle checkpoints (nsc_amd/tf_checkpoint.py): what the reference's Saver reads and writes (cmrl.py:64-72, nsc_module:548)
def test_crc32c_known_answers_and_mask():
    from nsc_amd.tf_checkpoint import crc32c, mask_crc, unmask_crc
    # RFC 3720 B.4 test vectors (the ones leveldb's crc32c_test holds too)
    assert crc32c(b"123456789") == 0xe3069283
    assert crc32c(bytes(32)) == 0x8a9136aa
    assert crc32c(b"\xff" * 32) == 0x62a8ab43
    assert crc32c(bytes(range(32))) == 0x46dd794e
    assert crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
    assert crc32c(b"hello world") == crc32c(b"world", crc32c(b"hello "))       # extendable
    for v in (0, 1, 0xe3069283, 0xffffffff):
        assert unmask_crc(mask_crc(v)) == v and mask_crc(v) != v


def test_tf_checkpoint_round_trip_names_shapes_dtypes(tmp_path):
    from nsc_amd.tf_checkpoint import write_checkpoint, read_checkpoint, read_index
    from tests._util import make_store
    named = dict(make_store(2, [[2], [2]], [32, 32], lpc=True).params)
    named["global_step"] = np.array(12345, np.int64)
    named["flags/on"] = np.array([True, False, True])
    named["half"] = np.arange(6, dtype=np.float64).reshape(2, 3)
    prefix = str(tmp_path / "model_bnn_ac_x_.ckpt")
    write_checkpoint(prefix, named)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    back = read_checkpoint(prefix)
    assert set(back) == set(named)
    for k, v in named.items():
        v = np.asarray(v)
        assert back[k].shape == v.shape and back[k].dtype == v.dtype, k
        np.testing.assert_array_equal(back[k], v)
    # a subset by name; the index alone lists every tensor, sorted (SSTable order), with offsets that tile the data shard
    sub = read_checkpoint(prefix, names=["scope_1/alpha", "global_step"])
    assert set(sub) == {"scope_1/alpha", "global_step"}
    from nsc_amd.tf_checkpoint import _parse_entry
    table = read_index(prefix + ".index")
    keys = list(table)
    assert keys == sorted(keys) and keys[0] == b"" and table[b""][:2] == b"\x08\x01"     # BundleHeaderProto.num_shards = 1
    end = 0
    for e in sorted((_parse_entry(v) for k, v in table.items() if k), key=lambda e: e["offset"]):
        assert e["offset"] == end and e["shard_id"] == 0
        end += e["size"]
    assert end == os.path.getsize(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    import struct
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57                        # table/format.cc kTableMagicNumber


def test_tf_checkpoint_many_blocks_and_prefix_compression(tmp_path):
    from nsc_amd.tf_checkpoint import write_index, read_index
    # 400 keys with long shared prefixes in 256-byte blocks: many data blocks, a multi-entry index block, restarts every 16 keys
    items = {f"scope_{i % 3 + 1}/conv1d_{i}/kernel".encode(): bytes([i % 251]) * (i % 7 + 1) for i in range(400)}
    path = str(tmp_path / "t.index")
    write_index(path, items, block_bytes=256)
    got = read_index(path)
    assert got == items and list(got) == sorted(items)
    raw = open(path, "rb").read()
    payload = sum(len(k) + len(v) for k, v in items.items())
    assert len(raw) < payload + 400 * 3                               # shared prefixes were elided (3 varints per entry + blocks' trailers otherwise)
    # the same table assembled by hand, one entry per block with NO prefix sharing, reads the same: the reader does not lean on the writer's choices
    from nsc_amd.tf_checkpoint import _emit_block, _block, TABLE_MAGIC
    import struct
    hand = str(tmp_path / "h.index")
    with open(hand, "wb") as f:
        index = [(k, _emit_block(f, _block([(k, items[k])]))) for k in sorted(items)[:40]]
        meta = _emit_block(f, _block([]))
        idx = _emit_block(f, _block(index, restart_interval=1))
        f.write((meta + idx).ljust(40, b"\x00") + struct.pack("<Q", TABLE_MAGIC))
    assert read_index(hand) == {k: items[k] for k in sorted(items)[:40]}


def test_tf_checkpoint_rejects_corruption(tmp_path):
    from nsc_amd.tf_checkpoint import write_checkpoint, read_checkpoint
    prefix = str(tmp_path / "c.ckpt")
    write_checkpoint(prefix, {"a/kernel": np.arange(12, dtype=np.float32).reshape(3, 4), "a/bias": np.ones(4, np.float32)})
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[5] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        read_checkpoint(prefix)
    write_checkpoint(prefix, {"a/kernel": np.zeros((3, 4), np.float32)})
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01                                                   # inside the first data block
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError):
        read_checkpoint(prefix)


def test_trainer_restores_from_a_tf_format_checkpoint(tmp_path):
    """restore() falls back to <prefix>.index when the .npz is absent - the path a user arriving with the reference's checkpoints
    takes (cmrl.py:67 saver.restore(sess, './check/model_bnn_ac_<id>_.ckpt'))."""
    from nsc_amd.tf_checkpoint import write_checkpoint
    from nsc_amd.neural_speech_coding_module import neuralSpeechCodingModule as NeuralSpeechCoding
    from tests._util import make_store
    named = dict(make_store(2, [[2], [2]], [32, 32], lpc=False).params)
    m = NeuralSpeechCoding.__new__(NeuralSpeechCoding)
    m._out_root, m._rand_model_id = str(tmp_path), "77"
    os.makedirs(tmp_path / "check")
    write_checkpoint(m.ckpt_path("")[:-len(".npz")], named)

    class Eng:
        def load_named(self, named):
            self.got = named
    e = Eng()
    m.restore(e, "", scopes=["scope_1"])                             # cmrl.py:332-347: a follower stage restores the earlier scopes only
    want = {k for k in named if k.startswith("scope_1/")}
    assert set(e.got) == want and 0 < len(want) < len(named)
    for k in want:
        np.testing.assert_array_equal(e.got[k], np.asarray(named[k]))
    m.restore(e, "")
    assert set(e.got) == set(named)


# ---- op surface host logic (nsc_amd/ops.py): the index maps of the per-store image set, on CPU ----
def test_flip_index_is_the_flipped_transposed_kernel():
    from nsc_amd.ops import _flip_index
    for K, Cin, Cout in ((9, 20, 100), (15, 20, 20), (1, 100, 20), (55, 1, 100)):
        w = np.arange(K * Cin * Cout, dtype=np.int64).reshape(K, Cin, Cout) + 1000
        wt = w[::-1].transpose(0, 2, 1)                                   # wt[k', o, i] = w[K - 1 - k', i, o]
        assert np.array_equal(_flip_index(K, Cin, Cout, 1000), wt.reshape(-1))


@pytest.mark.parametrize("C_,Cin,dil", [(100, 100, 1), (50, 50, 2), (100, 1, 1)])
def test_block_image_pair_index_composes_the_flip_with_the_data_gradient_image(C_, Cin, dil):
    """ops._block_image_index: [forward image | data-gradient image] of a gated block straight from its eight parameters.  The second
    part must equal the library's own data-gradient image index applied to the FLIPPED kernels (what the engine gathers in two
    steps); the first part the library's forward (split) image index; both from the same offsets."""
    import ctypes as C
    from nsc_amd import _lib, ops
    lib = _lib.load()
    sizes = [Cin * 20, 20, 6000, 20, 6000, 20, 9 * 20 * C_, C_]
    offs = tuple(int(v) for v in (np.concatenate([[0], np.cumsum([s_ + 4 for s_ in sizes])[:-1]]) + 64))    # (gaps: any placement works)
    meta = ops._block_image_meta(lib, C_, Cin, dil)
    assert meta is not None
    nf, kind, nb = meta
    idx = ops._block_image_index(lib, C_, Cin, dil, offs)
    assert idx.shape == (nf + nb,) and idx.dtype == np.int32 and nf % 4 == 0
    # forward part == the library's index for the same offsets
    n = int(lib.nsc_gated_block_simage_words(0, C_, Cin, dil)) if kind == "split" else int(lib.nsc_gated_block_image_floats(0, C_, Cin, dil))
    ref = np.empty(n, np.int32)
    fn = lib.nsc_gated_block_simage_index if kind == "split" else lib.nsc_gated_block_image_index
    assert fn(0, C_, Cin, dil, (C.c_long * 8)(*offs), ref.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(idx[:n], ref) and np.all(idx[n:nf] == -1)
    # data-gradient part: gather it from a parameter buffer == the library's image of the flipped kernels
    rng = np.random.default_rng(0)
    buf = rng.standard_normal(offs[-1] + sizes[-1] + 8).astype(np.float32)
    w1 = buf[offs[0]:offs[0] + sizes[0]].reshape(1, Cin, 20)
    wl = buf[offs[2]:offs[2] + sizes[2]].reshape(15, 20, 20)
    wr = buf[offs[4]:offs[4] + sizes[4]].reshape(15, 20, 20)
    w9 = buf[offs[6]:offs[6] + sizes[6]].reshape(9, 20, C_)
    flipped = np.concatenate([w[::-1].transpose(0, 2, 1).reshape(-1) for w in (w1, wl, wr, w9)])
    n1, n15 = Cin * 20, 6000
    bw = np.empty(nb, np.int32)
    assert lib.nsc_gated_block_image_index(1, C_, Cin, dil, (C.c_long * 4)(0, n1, n1 + n15, n1 + 2 * n15), bw.ctypes.data_as(C.c_void_p)) == 0
    want = np.where(bw >= 0, flipped[np.maximum(bw, 0)], 0.0)
    got_i = idx[nf:]
    assert np.all(got_i < (1 << 26))                                      # (plain words: no split modes in the exact image)
    got = np.where(got_i >= 0, buf[np.maximum(got_i, 0)], 0.0)
    assert np.array_equal(got, want)
