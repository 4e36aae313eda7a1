#!/bin/bash
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_reference_exec_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "recon_loss or loss or headline or alpha_minus or golden or phases or match_reference" 2>&1 | tail -6
