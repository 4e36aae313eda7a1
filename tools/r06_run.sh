#!/bin/bash
mkdir -p gpurun_out/r06i
timeout 600 python tools/dgrad3l_time.py > gpurun_out/r06i/dgrad3l_time.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06i/dgrad3l_time.txt
