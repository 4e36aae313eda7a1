cd $GRAFT_REPO_ROOT
git log -1 --format=%H 2>/dev/null | head -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r06ah_bench_default.json 2> gpurun_out/r06ah_bench_default.err; tail -c 1500 gpurun_out/r06ah_bench_default.json | head -c 600; echo
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r06ah_bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["op_surface"]["frac_of_engine"], d["op_surface"]["ms_per_step"], d["op_surface"]["engine_config2_ms_per_step"], d["codec_forward"]["us_per_frame"])
PY
python bench.py --config 2 --no-cpu-baseline --no-infer --no-op-surface 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo " (config 2)"
python bench.py --follower --no-cpu-baseline --no-infer --no-op-surface 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '; echo " (follower)"
