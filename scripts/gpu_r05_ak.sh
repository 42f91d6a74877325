#!/bin/bash
# round 5, call AK: the shader's workgroup size = rays per slot request of its fused compaction (one returning atomic on one word per workgroup): 256 / 512 / 1024
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -3
( for rep in 1 2; do for b in 256 512 1024; do
    echo "== RODENT_HIP_SHADE_BLOCK=$b rep $rep"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --spp 64
  done; done
  for b in 256 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b gallery"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --scene gallery --spp 16; done
  for b in 256 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b cornell, streaming"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 --mapping streaming; done
  for b in 256 1024; do echo "== RODENT_HIP_SHADE_BLOCK=$b 256 spp"; RODENT_HIP_SHADE_BLOCK=$b timeout 600 python scripts/frame_rate.py --spp 256; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/shade_block.txt; cat gpurun_out/r05/shade_block.txt
