#!/bin/bash
# round 5, call AF: the one-chunk / persistent switch point (384 Ki rays, fitted on the atrium) on the three other scene classes; the two-rank bench.py test after the weak / strong swap
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1500 python scripts/threshold_sweep.py --scenes gallery,crown,plant 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/threshold_sweep_classes.txt; cat gpurun_out/r05/threshold_sweep_classes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_ranks or bench_py_contract" 2>&1 | tail -3
