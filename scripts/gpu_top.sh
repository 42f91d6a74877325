#!/bin/bash
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
RODENT_HIP_LAB=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep or stream or overflow or bijection or special" 2>&1 | tail -4
RODENT_HIP_LAB=1 timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --big --only "$1" 2>&1 | tee gpurun_out/r02/sweep_top.log | cut -c1-200
