#!/bin/bash
# usage: bash tools/pmc_run.sh <tag>   (collect PMC passes of a short eager bench; writes gpurun_out/pmc_<tag>/passN)
# Counters go in separate passes with nothing but --pmc (no trace domains), as MI355X_MICROARCH.md prescribes.
TAG=${1:-x}; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --prof-steps 0 --no-infer --no-overlap"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pass1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pass2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pass3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pass4 -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 $GRAFT_REPO_ROOT/tools/pmc_calib.py > $OUT/c1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 $GRAFT_REPO_ROOT/tools/pmc_calib.py > $OUT/c2.log 2>&1
cd $GRAFT_REPO_ROOT
for p in pass1 pass2 pass3 pass4 calib_fetch calib_write; do python3 tools/pmc_summary.py $OUT/$p 3 > $OUT/summary_$p.txt 2>&1; done
python3 tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
find $OUT -name "*counter_collection.csv" -size +20M -delete
ls $OUT
