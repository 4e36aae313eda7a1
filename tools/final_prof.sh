#!/bin/bash
# round 6: profiles of the final code (bench line, rocprofv3 kernel stats default + isolated, PMC passes, one graph-replayed step, other configs)
TAG=${1:-r06p}
bash tools/prof_run.sh $TAG > gpurun_out/prof_${TAG}_stdout.txt 2>&1
bash tools/pmc_run.sh $TAG > gpurun_out/pmc_${TAG}_stdout.txt 2>&1
mkdir -p gpurun_out/cfg_$TAG
python bench.py --config 2 --no-cpu-baseline > gpurun_out/cfg_$TAG/bench_config2.json 2> gpurun_out/cfg_$TAG/c2.err
python bench.py --config 4 --no-cpu-baseline --no-infer --no-op-surface > gpurun_out/cfg_$TAG/bench_config4.json 2> gpurun_out/cfg_$TAG/c4.err
python bench.py --follower --no-cpu-baseline --no-infer --no-op-surface > gpurun_out/cfg_$TAG/bench_config3_follower.json 2> gpurun_out/cfg_$TAG/f.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cfg_$TAG/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-op-surface --passes 1 > $GRAFT_REPO_ROOT/gpurun_out/cfg_$TAG/tl.json 2> $GRAFT_REPO_ROOT/gpurun_out/cfg_$TAG/tl.err
cd $GRAFT_REPO_ROOT
t=$(find gpurun_out/cfg_$TAG/tl -name "*kernel_trace.csv" | head -1)
python3 tools/graph_step_timeline.py $t > gpurun_out/cfg_$TAG/step_timeline_graph.txt 2>&1
find gpurun_out/cfg_$TAG/tl -name "*kernel_trace.csv" -size +30M -delete
tail -3 gpurun_out/cfg_$TAG/step_timeline_graph.txt
grep "^{" gpurun_out/prof_$TAG/bench_default.json | cut -c1-300
