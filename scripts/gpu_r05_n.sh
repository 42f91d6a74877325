#!/bin/bash
# round 5, call N: the child-id load of a step by node lanes only (a triangle lane needs its 48 bytes and nothing else) against the build before it
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/chmask.txt; rm -f $O
for rep in 1 2 3; do
  for lib in exp_base librodent_hip; do
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
for rep in 1 2; do
  for lib in exp_base librodent_hip; do
    echo "== $lib" >> $O
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O
  done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "benchmark_rays_bit_exact or cornell_golden or deep_stack or special_tmin or triangle_soups or scene_classes or film_matches or every_bsdf" 2>&1 | tail -3 >> $O
cat $O
