#!/bin/bash
# usage: bash tools/prof_run.sh <tag>   kernel-trace + stats of (a) the default bench, (b) the isolated (no overlap, no inference) bench
TAG=${1:-x}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $R/bench.py > $OUT/bench_default_under_rocprof.json 2> $OUT/default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/isolated -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph --no-infer --no-overlap > $OUT/bench_isolated_under_rocprof.json 2> $OUT/isolated.err
cd $R
for d in default isolated; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv
  t=$(find $OUT/$d -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $t > $OUT/${d}_trace_by_grid.txt 2>&1
  find $OUT/$d -name "*kernel_trace.csv" -size +30M -delete
done
cat $OUT/bench_default.json
head -40 $OUT/isolated_trace_by_grid.txt
