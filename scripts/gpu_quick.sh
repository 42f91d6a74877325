#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
RODENT_HIP_LAB=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not bench" 2>&1 | tail -3
bash scripts/gpu_r03_profiles_traversal.sh r03
