#!/bin/bash
# soak: the subprocess / two-rank / two-process tests five times over
for i in 1 2 3 4 5; do timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_surface_gpu.py -m gpu -q -x -k "two_ranks or two_processes or cli_two" 2>&1 | tail -1; done
