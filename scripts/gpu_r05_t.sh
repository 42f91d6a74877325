#!/bin/bash
# round 5, call T: pixels generated block by block (RODENT_HIP_PIXEL_BLOCK) -- config 5's frame at 256 spp and at 64 spp, a film check
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/pixel_blocks.txt; rm -f $O
for rep in 1 2; do for b in 0 8 16 32 64; do
  echo "== RODENT_HIP_PIXEL_BLOCK=$b rep $rep" >> $O
  RODENT_HIP_PIXEL_BLOCK=$b timeout 600 python scripts/frame_rate.py --spp 256 --frames 2 2>&1 | tail -1 >> $O
done; done
for b in 0 16; do echo "== 64 spp RODENT_HIP_PIXEL_BLOCK=$b" >> $O; RODENT_HIP_PIXEL_BLOCK=$b timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O; done
cat $O
RODENT_HIP_PIXEL_BLOCK=16 timeout 900 python -m pytest tests/test_gpu_render.py -q -x 2>&1 | tail -3
