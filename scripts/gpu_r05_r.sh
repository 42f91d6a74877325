#!/bin/bash
# round 5, call R2: tile mapping in the one-chunk kernel too -- test, switch-point sweep with it on / off
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "image_order" 2>&1 | tail -5
for g in 0 -1; do echo "== RODENT_HIP_RAY_GRID=$g"; RODENT_HIP_RAY_GRID=$g timeout 900 python scripts/threshold_sweep.py --scenes atrium,cornell 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/threshold_sweep_grid.txt
