#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
timeout 900 python scripts/refill_sweep.py --idle 0,32,40,48 --scenes atrium,atrium/2,atrium/8,atrium/16 --frames 3 > gpurun_out/r03/refill_sweep_32Mi.txt 2>&1; tail -5 gpurun_out/r03/refill_sweep_32Mi.txt | cut -c1-110
bash scripts/gpu_r03_profiles_render.sh r03
