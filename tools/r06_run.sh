#!/bin/bash
mkdir -p gpurun_out/r06n
B="python bench.py --no-cpu-baseline --no-infer --no-op-surface --steps 20 --warmup 5 --passes 3"
for t in a b; do
$B > gpurun_out/r06n/on_$t.json 2> gpurun_out/r06n/on_$t.err
NSC_DEFER_SMALL=0 $B > gpurun_out/r06n/off_$t.json 2> gpurun_out/r06n/off_$t.err
NSC_DEFER_SMALL=0 NSC_LIVE_GATHER=0 $B > gpurun_out/r06n/off2_$t.json 2> gpurun_out/r06n/off2_$t.err
done
for f in on_a off_a off2_a on_b off_b off2_b; do echo -n "$f: "; python - <<PY
import json
l=[x for x in open("gpurun_out/r06n/$f.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["ms_per_step"], d["value"])
else:
    print(open("gpurun_out/r06n/$f.err").read()[-1500:])
PY
done
