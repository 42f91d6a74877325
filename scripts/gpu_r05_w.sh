#!/bin/bash
# round 5, call W: the second reading of the probes (per-pixel lists that are no camera dump) -- test, scene matrix, the benchmark sets with recognition on / off
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "image_order" 2>&1 | tail -15
for g in 0 -1 0 -1; do RODENT_HIP_RAY_GRID=$g timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids | sed "s/^/grid $g: /"; done | tee gpurun_out/r05/grid_onoff_sets.txt
bash scripts/gpu_r05_scenes.sh > gpurun_out/r05/scenes.log 2>&1; cat gpurun_out/r05/scene_matrix.txt | cut -c1-400
