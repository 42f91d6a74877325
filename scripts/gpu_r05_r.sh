#!/bin/bash
# round 5, call R: tile mapping with detection -- its test, the suite, on / off on three scenes
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "image_order or switches_kernels or chooses_chunks" 2>&1 | tail -5
for s in atrium gallery crown; do for g in 0 -1; do RODENT_HIP_RAY_GRID=$g timeout 600 python scripts/grid_experiment.py $s 1024 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r05/grid_detect.txt
timeout 600 python scripts/fixed_costs.py 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/r05/fixed_costs_after.txt
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05/tests_r.txt 2>&1; tail -6 gpurun_out/r05/tests_r.txt
