#!/bin/bash
# Sweeps the late-block fraction / priority of the fast kernel (RODENT_HIP_LATE_FRAC, RODENT_HIP_LATE_PRIO).
for cfg in "1.0 0" "0.9 0" "0.8 0" "0.7 0" "0.6 0" "0.5 0" "0.3 0" "0.0 0" "0.8 1" "0.6 1" "0.5 1"; do
  set -- $cfg
  echo "== late_frac $1 prio $2"
  RODENT_HIP_LATE_FRAC=$1 RODENT_HIP_LATE_PRIO=$2 timeout 120 python scripts/sweep.py --steps 30 --variants 0 2>&1 | tail -1
done
echo "== reversed ray order (late off)"; timeout 120 python scripts/sweep.py --steps 30 --variants 0,3 --reverse 2>&1 | tail -2
