#!/bin/bash
# round 5, call AL: the shader's workgroup size, second sweep (512 on the other scenes, 256 spp)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
( for b in 256 512 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b 256 spp"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --spp 256; done
  for b in 256 512 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b gallery"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --scene gallery --spp 16; done
  for b in 256 512 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b cornell, streaming"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 --mapping streaming; done
  for b in 256 512 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b crown"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --scene crown --spp 16; done ) 2>&1 | grep -v "amdgpu.ids\|Missing material" > gpurun_out/r05/shade_block2.txt; cat gpurun_out/r05/shade_block2.txt
