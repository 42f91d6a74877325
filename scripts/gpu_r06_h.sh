#!/bin/bash
# round 6, call H: triangle turns with deferral (lab variants turns-p*): no rounds of their own, the triangle half in every P-th iteration
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
RODENT_HIP_LAB=1 timeout 1200 python scripts/defer_experiment.py --only 'turns-p' --big --steps 30 2>&1 | grep -v "amdgpu.ids\|stats-defer" | tee gpurun_out/r06/turns_experiment.txt
