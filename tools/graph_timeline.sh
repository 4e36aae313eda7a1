#!/bin/bash
# one hipGraph-replayed step of the default bench in launch order (rocprofv3 kernel trace; 300 replayed steps so that the median span is a replay)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python3 $R/bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-infer --no-op-surface --passes 1 > $O/bench.json 2> $O/err.txt
cd $R
t=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/graph_step_timeline.py $t > gpurun_out/step_timeline_graph.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat gpurun_out/step_timeline_graph.txt
