#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
timeout 900 python scripts/refill_sweep.py --idle 0,40 --scenes atrium,atrium/2,atrium/8 --frames 3 2>&1 | tail -4 | cut -c1-110
timeout 900 python scripts/refill_sweep.py --idle 40 --scenes atrium --size 3840x2160 --spp 32 --frames 2 2>&1 | tail -1 | cut -c1-110
