#!/bin/bash
# round 5, call AH: after the change to render.hip (k_trace_refill on slabs, base + offset addressing): the whole -m gpu suite, then the renderer's profiles and bench.py
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_ah.txt 2>&1; tail -3 gpurun_out/r05/tests_ah.txt
bash scripts/gpu_r03_profiles_render.sh r05
