#!/bin/bash
mkdir -p gpurun_out/r06j
B="python bench.py --no-cpu-baseline --no-infer --no-op-surface --steps 20 --warmup 5 --passes 3"
$B > gpurun_out/r06j/bench_default.json 2> gpurun_out/r06j/bench_default.err
NSC_PAIR_BWD=0 $B > gpurun_out/r06j/bench_nopairbwd.json 2> gpurun_out/r06j/bench_nopairbwd.err
NSC_SPLIT_DGRAD=1 $B > gpurun_out/r06j/bench_splitdgrad.json 2> gpurun_out/r06j/bench_splitdgrad.err
$B > gpurun_out/r06j/bench_default2.json 2> gpurun_out/r06j/bench_default2.err
for f in default nopairbwd splitdgrad default2; do echo $f; python - <<PY
import json
l=[x for x in open("gpurun_out/r06j/bench_$f.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["ms_per_step"], d.get("kernels",{}).get("block_dgrad"), d.get("roofline"))
else:
    print(open("gpurun_out/r06j/bench_$f.err").read()[-1500:])
PY
done
