#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
RODENT_HIP_LAB=1 python scripts/order_experiment.py --cameras > gpurun_out/r03/order_experiment.txt 2>&1; tail -50 gpurun_out/r03/order_experiment.txt
