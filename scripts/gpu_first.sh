#!/bin/bash
# first GPU call of a round: the whole -m gpu suite, the VALU issue-rate microbenchmark, the layout sweep
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu.log; tail -5 gpurun_out/r02/pytest_gpu.log
timeout 120 rodent_amd/bin/valu_peak > gpurun_out/r02/valu_peak.txt 2>&1; tail -12 gpurun_out/r02/valu_peak.txt
timeout 600 python scripts/sweep_widths.py --big --all-variants > gpurun_out/r02/sweep_widths.log 2>&1; cat gpurun_out/r02/sweep_widths.log
