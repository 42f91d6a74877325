#!/bin/bash
# round 5, call Y: the whole suite and the round's evidence on the final kernels
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05/tests_final.txt 2>&1; tail -4 gpurun_out/r05/tests_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python scripts/range_costs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/range_costs.txt; head -8 gpurun_out/r05/range_costs.txt
bash scripts/gpu_r05_profiles.sh r05
