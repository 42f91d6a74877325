#!/bin/bash
# lab evidence of the top-image family: variant sweep, chunk-order experiment, chunk timeline
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
RODENT_HIP_LAB=1 timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --big --only top15,top31w2,top63w4,top127w8,top255w16,top23,top15-keep,top127w8-keep,sorted-top63w4,top255p16-pf,top127p8,top63p4,top15p1,sorted-top255p16,top255p16-o16,top1023p16-o16,top255p8-o24,top63p4-o28,top255r16-32,top255r16-48,top-fused,top-prio64,top-prio96,top-prio128,fast,phased,sorted 2>&1 | tee gpurun_out/r02/sweep_top_family.log | cut -c1-130
RODENT_HIP_LAB=1 timeout 300 python scripts/lpt_experiment.py 2>&1 | tee gpurun_out/r02/lpt_experiment.txt | tail -9
RODENT_HIP_LAB=1 timeout 300 python scripts/trace_top.py 2>&1 | tee gpurun_out/r02/trace_top.txt | tail -3
