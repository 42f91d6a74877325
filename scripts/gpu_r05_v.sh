#!/bin/bash
# round 5, call V: tile order -- consecutive tiles run down a stack of 1 / 2 / 4 / 8 / 16 bands before they move right (a 2048-position group = 256x8, 128x16, 64x32, 32x64, 16x128 pixels)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for s in atrium gallery crown; do for rep in 1 2; do for sb in 1 2 4 8 16; do
  RODENT_HIP_LIB=rodent_amd/lib/exp_stack.so RODENT_HIP_RAY_GRID=$((1024 + sb * 1048576)) timeout 600 python scripts/grid_experiment.py $s 1024 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/stack $sb: /"
done; done; done | tee gpurun_out/r05/tile_stacks.txt
