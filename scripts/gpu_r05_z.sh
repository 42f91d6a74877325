#!/bin/bash
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "schedule_history_only" 2>&1 | tail -1; done
echo "== whole module"
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -1; done
echo "== grid off, whole module"
for i in 1 2 3; do RODENT_HIP_RAY_GRID=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -1; done
