#!/bin/bash
# per-kernel GPU time of the op-surface step (tools/op_surface_prof.py: 23 steps of forward + loss + backward through nn_core_operator)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/surf
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/surf -o surf -- python3 $R/tools/op_surface_prof.py 1 > $R/gpurun_out/surf/run.log 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/surf/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU time per step (23 steps): %.1f us in %d dispatches" % (tot/23/1e3, sum(int(r["Calls"]) for r in rows)/23))
for r in rows[:50]:
    print("%-100s %6.1f calls/step %8.1f us/step %7.1f us/call" % (r["Name"][:100], int(r["Calls"])/23, float(r["TotalDurationNs"])/23/1e3, float(r["AverageNs"])/1e3))
PY
grep "steps:" gpurun_out/surf/run.log
find gpurun_out/surf -name "*kernel_trace.csv" -delete
