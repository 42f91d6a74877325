#!/bin/bash
# Sweeps the scheduling knobs of the fast kernel's single-step loop (RODENT_HIP_LATE_FRAC, RODENT_HIP_TRI_MIN).
for cfg in "0.0 1" "0.0 2" "0.0 4" "0.0 8" "0.0 16" "0.0 32"; do
  set -- $cfg
  echo "== late_frac $1 tri_min $2"
  RODENT_HIP_LATE_FRAC=$1 RODENT_HIP_TRI_MIN=$2 timeout 120 python scripts/sweep.py --steps 30 --variants 0 2>&1 | tail -1
done
