#!/bin/bash
# round 5, call A: the in-lane stack spill against round 4's library (same box, back to back), the new deep-stack tests, then the whole -m gpu suite
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
R04=$PWD/rodent_amd/lib/librodent_hip_r04.so
for rep in 1 2; do
  echo "== r04 library, rep $rep" >> $O/ab_spill.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/sweep_auto.py --steps 40 --variants top,refill,fast 2>&1 | grep -v amdgpu.ids >> $O/ab_spill.txt
  echo "== r05 library, rep $rep" >> $O/ab_spill.txt
  timeout 600 python scripts/sweep_auto.py --steps 40 --variants top,refill,fast 2>&1 | grep -v amdgpu.ids >> $O/ab_spill.txt
done
for rep in 1 2; do
  echo "== r04 library, rep $rep" >> $O/ab_spill_render.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/ab_spill_render.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 2>&1 | tail -1 >> $O/ab_spill_render.txt
  echo "== r05 library, rep $rep" >> $O/ab_spill_render.txt
  timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/ab_spill_render.txt
  timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 2>&1 | tail -1 >> $O/ab_spill_render.txt
done
timeout 900 python -m pytest tests -m gpu -x -q -s -k "deep_stack or stack_overflow or several_streams" > $O/tests_deep.txt 2>&1
tail -5 $O/tests_deep.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -5 $O/tests_all.txt
cat $O/ab_spill.txt $O/ab_spill_render.txt
