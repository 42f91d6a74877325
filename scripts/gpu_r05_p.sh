#!/bin/bash
# round 5, call P: after the last kernel change -- suite, strong-scaling prediction, the round's evidence again (profiles are stamped with the source hash), bench
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_final.txt 2>&1; tail -4 gpurun_out/r05/tests_final.txt
timeout 600 python scripts/range_costs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/range_costs.txt; cat gpurun_out/r05/range_costs.txt
bash scripts/gpu_r05_profiles.sh r05
