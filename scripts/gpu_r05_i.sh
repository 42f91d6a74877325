#!/bin/bash
# round 5, call I: the auto kernel's refill loop as a function of its own
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/outline_refill.txt; rm -f $O
for rep in 1 2 3; do timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O; done
timeout 900 python scripts/scene_matrix.py --scenes atrium,crown 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python -m pytest tests -m gpu -x -q -k "benchmark_rays_bit_exact or default_mapping or deep_stack" 2>&1 | tail -3 >> $O
cat $O
