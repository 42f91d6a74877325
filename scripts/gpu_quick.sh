#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_refbuilt.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_atrium.py -x -q -m gpu -k "bit_exact or variant or cross" 2>&1 | tail -4
