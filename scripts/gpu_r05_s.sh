#!/bin/bash
# round 5, call S: after the tile mapping -- suite, strong-scaling prediction, the round's evidence (profiles are stamped with the source hash), the scene matrix
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05/tests_final.txt 2>&1; tail -4 gpurun_out/r05/tests_final.txt
timeout 600 python scripts/range_costs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/range_costs.txt; cat gpurun_out/r05/range_costs.txt
bash scripts/gpu_r05_profiles.sh r05
bash scripts/gpu_r05_scenes.sh
