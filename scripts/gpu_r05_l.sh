#!/bin/bash
# round 5, call L: what origin-cell order would be worth to the bounce passes on THIS round's kernels (VERDICT r4 item 2's exit clause)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1200 python scripts/bounce_coherence_experiment.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/bounce_coherence_experiment.txt
cat gpurun_out/r05/bounce_coherence_experiment.txt
