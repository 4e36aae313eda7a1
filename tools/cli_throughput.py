#!/usr/bin/env python3
"""CLI-level throughput: main.py over a synthetic --data_root file, feeder included.

The bench times a batch resident in HBM; the product trainer gathers rows from a memory-mapped file, copies them through two
pinned buffers (_Feeder) and launches the step from Python.  This script writes a synthetic frame file, runs

    main.py --training_mode 1   (one_ae: 1 codec, pretrain epoch without the quantizer, then quan + entropy epochs)

for EPOCHS epochs of STEPS steps at batch 128, and reports the frames/s the trainer itself prints per epoch (its timer covers
the training loop of the epoch only, like the reference's per-epoch elapsed: neural_speech_coding_module.py:452-460) next to
`bench.py --config 2` (the same step on a resident batch).  With --lpc it runs the LPC-domain (collaborative quantisation)
one_ae phase instead.  Output: one JSON line; keep it under profiles/.

    python tools/cli_throughput.py [--steps 250] [--epochs 4] [--batch 128] [--lpc] [--no-bench]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--lpc", action="store_true")
    ap.add_argument("--no-bench", action="store_true")
    ap.add_argument("--extra", type=str, default="", help="extra main.py flags, space separated")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="nsc_cli_")
    rng = np.random.default_rng(11)
    win = np.concatenate([np.hanning(63)[:32], np.ones(448), np.hanning(63)[31:]])
    n = a.batch * (a.steps + 2)
    frames = (np.clip(0.03 * rng.standard_normal((n, 512)), -1, 1) * win).astype(np.float32)
    if a.lpc:   # rows = frame | 16 LSFs | residual (the layout the LPC trainer reads)
        lsf = np.sort(rng.uniform(0.03, 3.1, (n, 16)), axis=1).astype(np.float32)
        frames = np.concatenate([frames, lsf, frames], axis=1)
    data = os.path.join(tmp, "frames.npy")
    np.save(data, frames)
    flags = ["--learning_rate_tanh", "2e-4", "--learning_rate_greedy_followers", "2e-5 2e-6", "--epoch_tanh", str(a.epochs),
             "--epoch_greedy_followers", "1 1", "--from_where_step", "2", "--batch_size", str(a.batch), "--num_resnets", "1",
             "--training_mode", "1", "--base_model_id", "", "--suffix", "cli", "--window_size", "512",
             "--bottleneck_kernel_and_dilation", "9 9 100 20 1 2", "--is_cq", "1" if a.lpc else "0", "--the_strides", "2",
             "--save_unique_mark", "", "--coeff_term", "60 10 10 0.3", "--res_scalar", "1.0", "--pretrain_step", "1",
             "--target_entropy", "2.2", "--num_bins_for_follower", "32", "--data_root", data, "--max_batches_per_epoch",
             str(a.steps), "--out_root", tmp, "--model_id", "7654321", "--seed", "3", "--tau_from_validation", "0"]
    if a.lpc:
        flags.append("--lpc_domain")
    flags += a.extra.split()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), *flags], capture_output=True, text=True, cwd=tmp, timeout=3000)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        sys.exit(r.returncode)
    fps = [float(m) for m in re.findall(r"^frames/s: ([0-9.]+)", r.stdout, flags=re.M)]
    out = dict(cli="main.py --training_mode 1" + (" --lpc_domain" if a.lpc else ""), batch=a.batch, steps_per_epoch=a.steps,
               frames_per_s_by_epoch=fps, note="epoch 0 = the no-quantizer op (pretrain_step 1) and includes first-step allocation; "
               "later epochs = the quan op")
    if fps:
        out["frames_per_s_quan_op_median"] = float(np.median(fps[1:])) if len(fps) > 1 else fps[0]
    if not a.no_bench and not a.lpc:
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "50", "--batch", str(a.batch)],
                           capture_output=True, text=True, timeout=3000)
        line = [l for l in b.stdout.splitlines() if l.startswith("{")]
        if line:
            j = json.loads(line[-1])
            out["bench_config2_frames_per_s"] = j["value"]
            out["bench_launch"] = j["config"]["launch"]
            if fps:
                out["cli_over_bench"] = round(out["frames_per_s_quan_op_median"] / j["value"], 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
