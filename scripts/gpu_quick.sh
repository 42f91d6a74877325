#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
RODENT_HIP_LAB=1 python scripts/sweep_widths.py --widths 2 --all-variants --big --steps 20 --only top,top-one,top-double > gpurun_out/r03/sweep_double.log 2>&1; tail -5 gpurun_out/r03/sweep_double.log | cut -c1-220
