#!/bin/bash
mkdir -p gpurun_out/r06w
timeout 900 python -m pytest tests/test_block_split_gpu.py -q -x -k "three_launch" 2>&1 | tail -4
timeout 600 python tools/dgrad3l_time.py > gpurun_out/r06w/dgrad3l_time.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06w/dgrad3l_time.txt
