#!/bin/bash
# round 5, call M: the scene matrix again (two rounds over the mappings, the better one counts) + L2 hit rates
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
timeout 2400 python scripts/scene_matrix.py --steps 30 --json $O/scene_matrix.json 2>&1 | grep -v amdgpu.ids > $O/scene_matrix.txt
cat $O/scene_matrix.txt
