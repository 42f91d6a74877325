#!/bin/bash
# round 3, call A (lab build): the new finish / miss-record variants of the persistent kernel, and the chunk-shape experiment
mkdir -p gpurun_out/r03; export TMPDIR=/tmp
RODENT_HIP_LAB=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_stack or special or ragged or bijection or golden or overflow" 2>&1 | tail -5
RODENT_HIP_LAB=1 timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --big --only fast,top-fused,top-one,top-lazy,top-lazy-one 2>&1 | tee gpurun_out/r03/sweep_one_lazy.log | cut -c1-150
RODENT_HIP_LAB=1 timeout 900 python scripts/tile_experiment.py --big 2>&1 | tee gpurun_out/r03/tile_experiment.txt
RODENT_HIP_LAB=1 timeout 600 python scripts/tile_experiment.py --kernel top-userperm-lazy-one 2>&1 | tee gpurun_out/r03/tile_experiment_lazy_one.txt
