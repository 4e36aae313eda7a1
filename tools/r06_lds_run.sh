#!/bin/bash
mkdir -p gpurun_out/r06f
timeout 900 python -m pytest tests/test_block_split_gpu.py -q -x -k "three_launch" 2>&1 | tail -15 > gpurun_out/r06f/three_launch_tests.txt
cat gpurun_out/r06f/three_launch_tests.txt
timeout 600 python tools/dgrad3l_time.py > gpurun_out/r06f/dgrad3l_time.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06f/dgrad3l_time.txt
O=gpurun_out/r06f/cout1_share_stress.txt
echo "== one process alone" > $O; python tools/cout1_share_stress.py 6000 >> $O 2>&1
echo "== two processes side by side" >> $O; python tools/cout1_share_stress.py 20000 >> $O 2>&1 & P=$!; python tools/cout1_share_stress.py 20000 > $O.b 2>&1; wait $P; cat $O.b >> $O
rm -f $O.b; grep -v amdgpu.ids $O
