#!/bin/bash
mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "live_gather" 2>&1 | tail -6
B="python bench.py --no-cpu-baseline --no-infer --no-op-surface --steps 20 --warmup 5 --passes 3"
$B > gpurun_out/r06l/bench_live.json 2> gpurun_out/r06l/bench_live.err
NSC_LIVE_GATHER=0 $B > gpurun_out/r06l/bench_full.json 2> gpurun_out/r06l/bench_full.err
$B --follower > gpurun_out/r06l/bench_live_follower.json 2> gpurun_out/r06l/bench_live_follower.err
NSC_LIVE_GATHER=0 $B --follower > gpurun_out/r06l/bench_full_follower.json 2> gpurun_out/r06l/bench_full_follower.err
for f in live full live_follower full_follower; do echo -n "$f: "; python - <<PY
import json
l=[x for x in open("gpurun_out/r06l/bench_$f.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["ms_per_step"], d["value"])
else:
    print(open("gpurun_out/r06l/bench_$f.err").read()[-1500:])
PY
done
