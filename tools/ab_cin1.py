"""A/B on one box: where the weight gradients run (side stream per conv / batched at the tail of the step)."""
import json, subprocess, sys
for rep in range(2):
    for extra in ([], ["--no-overlap"], ["--no-batch-conv-wgrad"], ["--no-overlap", "--no-batch-conv-wgrad"]):
        for flag in (1, 0):
            code = ("import sys, nsc_amd.engine as E; E.CascadeEngine.batch_cin1_wgrad=bool(%d); import bench; "
                    "sys.argv=['bench.py','--steps','40','--warmup','8','--no-cpu-baseline','--no-infer','--prof-steps','1']+%r; bench.main()" % (flag, extra))
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(flag, extra, json.loads(line[-1])["ms_per_step"] if line else out.stderr[-400:], flush=True)
