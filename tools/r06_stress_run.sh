#!/bin/bash
mkdir -p gpurun_out/r06b
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1"
timeout 900 python tools/dp_race_stress.py 300 > gpurun_out/r06b/stress_1proc.txt 2>&1
timeout 900 $TR --master-port 29661 tools/dp_race_stress.py 300 > gpurun_out/r06b/stress_2rank.txt 2>&1
NSC_STRESS_PAIRS=0 timeout 900 python tools/dp_race_stress.py 300 > gpurun_out/r06b/stress_1proc_nopairs.txt 2>&1
grep -h "glitch\|tail_overlap" gpurun_out/r06b/*.txt | cut -c1-400
