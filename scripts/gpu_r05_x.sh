#!/bin/bash
# round 5, call X: tiles in the BVH4 / BVH8 kernels -- test; the three layouts side by side with recognition off / on
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "image_order" 2>&1 | tail -5
for g in 0 -1; do echo "# RODENT_HIP_RAY_GRID=$g"; RODENT_HIP_RAY_GRID=$g timeout 600 python scripts/width_compare.py atrium 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/width_compare.txt
