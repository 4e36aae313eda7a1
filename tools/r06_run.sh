#!/bin/bash
mkdir -p gpurun_out/r06o
( echo "# pytest tests -m gpu -x -q on $(git rev-parse --short HEAD 2>/dev/null || echo working-tree) (round 6 checkpoint)"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06o/gputests.txt
tail -6 gpurun_out/r06o/gputests.txt
