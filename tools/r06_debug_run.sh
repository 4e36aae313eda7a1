#!/bin/bash
# round 6, first GPU call: root-cause of the red two-rank LPC test + the whole suite without -x
mkdir -p gpurun_out/r06a
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1"
NSC_TAIL_OVERLAP=1 timeout 600 $TR --master-port 29655 tools/dp_layout_debug.py > gpurun_out/r06a/dbg_tail1.txt 2>&1
NSC_TAIL_OVERLAP=0 timeout 600 $TR --master-port 29656 tools/dp_layout_debug.py > gpurun_out/r06a/dbg_tail0.txt 2>&1
NSC_DEBUG_LPC=0 NSC_TAIL_OVERLAP=1 timeout 600 $TR --master-port 29657 tools/dp_layout_debug.py > gpurun_out/r06a/dbg_nolpc.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06a/gputests_all.txt 2>&1
tail -5 gpurun_out/r06a/gputests_all.txt
