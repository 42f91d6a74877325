#!/bin/bash
# round 5, call AM: after tri_tex: renderer parity (the textured scenes included, and the old path in its own process), then the renderer's profiles + bench.py
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_r03_profiles_render.sh r05
