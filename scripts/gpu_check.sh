#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "threshold or history" 2>&1 | tail -3
timeout 600 python scripts/big_launch_check.py 2>&1 | tail -2
RODENT_HIP_SCHEDULE_HISTORY=1 timeout 600 python scripts/big_launch_check.py 2>&1 | tail -2
