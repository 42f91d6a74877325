#!/bin/bash
# round 5, call G: LDS-image fetches and memory fetches of a step issued together (inline asm, one wait) against the compiler's two branches
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/joint_loads.txt; rm -f $O
for rep in 1 2 3; do
  for lib in librodent_hip exp_joint; do
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
RODENT_HIP_LIB=$PWD/rodent_amd/lib/exp_joint.so timeout 900 python -m pytest tests -m gpu -x -q -k "benchmark_rays_bit_exact or cornell_golden or deep_stack or special_tmin or triangle_soups or chunk_mapping" 2>&1 | tail -3 >> $O
echo "== librodent_hip" >> $O
timeout 900 python scripts/scene_matrix.py --scenes crown --variants top,fast 2>&1 | grep -v amdgpu.ids >> $O
echo "== exp_joint" >> $O
RODENT_HIP_LIB=$PWD/rodent_amd/lib/exp_joint.so timeout 900 python scripts/scene_matrix.py --scenes crown --variants top,fast 2>&1 | grep -v amdgpu.ids >> $O
cat $O
