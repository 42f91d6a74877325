#!/bin/bash
# round 5, call E: where the spill's cost comes from -- four builds of traversal.hip (reload test at the start / at the end of the step; error word in pinned host memory / in device memory) against round 4's library
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/spill_experiment.txt; rm -f $O
for rep in 1 2 3; do
  for lib in librodent_hip_r04 exp_start_host exp_start_dev exp_end_dev exp_end_host; do
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
