#!/bin/bash
# usage: bash tools/pmc_run.sh <tag>   (collect PMC passes of a short eager bench; writes gpurun_out/pmc_<tag>/passN)
TAG=${1:-x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --prof-steps 0 --no-infer --no-overlap"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pass1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pass2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pass3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pass4 -- $CMD > $OUT/p4.log 2>&1
cd $GRAFT_REPO_ROOT
for p in 1 2 3 4; do python3 tools/pmc_summary.py $OUT/pass$p 3 > $OUT/summary_pass$p.txt 2>&1; done
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls $OUT
