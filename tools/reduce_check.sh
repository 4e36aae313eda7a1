cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/red
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/red -o red -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-infer --no-op-surface --no-graph --no-overlap > $R/gpurun_out/red/bench.json 2> $R/gpurun_out/red/err.txt
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/red/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "reduce" in r["Name"] or "wgrad" in r["Name"]:
        print("%-90s %6d calls %8.1f us/call" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"])/1e3))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/red/bench.json | head -1
find gpurun_out/red -name "*kernel_trace.csv" -delete
