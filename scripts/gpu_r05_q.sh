#!/bin/bash
# round 5, call Q3: fixed cost of the persistent kernel without the per-workgroup statistics atomics / with the first rays' loads issued before the image is staged
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for lib in librodent_hip exp_nostats exp_early; do
  echo "== $lib"; RODENT_HIP_LIB=rodent_amd/lib/$lib.so FIXED_COSTS_VARIANTS=top,refill,fast timeout 600 python scripts/fixed_costs.py 2>&1 | grep -v amdgpu.ids | head -8
  for g in 0 1024; do RODENT_HIP_LIB=rodent_amd/lib/$lib.so RODENT_HIP_RAY_GRID=$g timeout 300 python scripts/grid_experiment.py atrium 1024 2>&1 | grep -v amdgpu.ids | head -1; done
done | tee gpurun_out/r05/fixed_costs_exp.txt
