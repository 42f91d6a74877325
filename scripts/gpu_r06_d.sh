#!/bin/bash
# round 6, call D: the cooperative BVH8 any-hit kernel (lab) and what child prefetch does to the small launches of a strong-scaling run
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
OUT=gpurun_out/r06
RODENT_HIP_LAB=1 timeout 900 python scripts/coop8_experiment.py 2>&1 | grep -v amdgpu.ids | tee $OUT/coop8_experiment.txt
RODENT_HIP_LAB=1 timeout 900 python scripts/range_costs.py --variants top,fast-pf0,fast-pf48 --worlds 8 2>&1 | grep -v amdgpu.ids | tee $OUT/range_costs_prefetch.txt
