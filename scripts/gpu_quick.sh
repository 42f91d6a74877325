#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
RODENT_HIP_LAB=1 python scripts/sweep_widths.py --widths 2 --all-variants --big-random --steps 20 --only top,refill,top255r16-16,top255r16-24,top255r16-40,top255r16-48 > gpurun_out/r03/sweep_refill_thresholds.log 2>&1; tail -8 gpurun_out/r03/sweep_refill_thresholds.log | cut -c1-200
