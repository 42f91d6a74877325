#!/bin/bash
# round 5, call B: the reworked spill helpers against round 4's library again; what the host pays around a synchronous launch
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
R04=$PWD/rodent_amd/lib/librodent_hip_r04.so
rm -f $O/ab_spill2.txt
for rep in 1 2 3; do
  echo "== r04 library, rep $rep" >> $O/ab_spill2.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/sweep_auto.py --steps 40 --variants top,refill 2>&1 | grep -v amdgpu.ids >> $O/ab_spill2.txt
  echo "== r05 library, rep $rep" >> $O/ab_spill2.txt
  timeout 600 python scripts/sweep_auto.py --steps 40 --variants top,refill 2>&1 | grep -v amdgpu.ids >> $O/ab_spill2.txt
done
echo "== r04 library" > $O/host_call_costs.txt
RODENT_HIP_LIB=$R04 timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
echo "== r05 library" >> $O/host_call_costs.txt
timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
timeout 900 python -m pytest tests -m gpu -x -q -s -k "deep_stack or stack_overflow or interleaved_row_tiles" > $O/tests_deep2.txt 2>&1
tail -4 $O/tests_deep2.txt
cat $O/ab_spill2.txt $O/host_call_costs.txt
