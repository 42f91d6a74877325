#!/bin/bash
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu2.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu2.log; tail -15 gpurun_out/r02/pytest_gpu2.log
timeout 900 python scripts/sweep_widths.py --big --all-variants --widths 2 > gpurun_out/r02/sweep_phased.log 2>&1; cat gpurun_out/r02/sweep_phased.log
