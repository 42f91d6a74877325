#!/bin/bash
# round 5, call K: the whole suite, then the round's evidence (scripts/gpu_r05_profiles.sh: kernel traces, HBM traffic, twelve PMC groups, renderer profiles, bench.py)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_all.txt 2>&1; tail -5 gpurun_out/r05/tests_all.txt
bash scripts/gpu_r05_profiles.sh r05
