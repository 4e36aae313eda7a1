#!/bin/bash
mkdir -p gpurun_out/r06z
( echo "# pytest tests -m gpu -x -q on commit 8f8ac4d (round 6, final code)"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r06z/gputests_final.txt
tail -4 gpurun_out/r06z/gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r06z/bench_default.json 2> gpurun_out/r06z/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06z/bench_default.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d.get('ms_per_step_exact_f32'), d['roofline']['frac'])"
