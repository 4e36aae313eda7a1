#!/bin/bash
mkdir -p gpurun_out/r06u
( echo "# pytest tests -m gpu -x -q: library built WITHOUT packed fp32 ops (round 6)"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06u/gputests.txt
tail -5 gpurun_out/r06u/gputests.txt
python bench.py > gpurun_out/r06u/bench_default.json 2> gpurun_out/r06u/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06u/bench_default.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d.get('ms_per_step_exact_f32'), d['roofline']['frac'], d['codec_forward'])"
