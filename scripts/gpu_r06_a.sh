#!/bin/bash
# round 6, call A: the ADVICE r5 fixes under the whole GPU suite, then the first deferred-leaf sweep (lab build)
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
RODENT_HIP_LAB=1 timeout 900 python scripts/defer_experiment.py --big 2>&1 | tee gpurun_out/r06/defer_experiment_a.txt | tail -30
