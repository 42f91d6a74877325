#!/bin/bash
# round 5, call AD: the renderer's refill thresholds on all four scene classes
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 2400 python scripts/refill_rule_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/refill_rule_check.txt; cat gpurun_out/r05/refill_rule_check.txt
