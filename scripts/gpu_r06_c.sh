#!/bin/bash
# round 6, call C: the miss record stored late: traversal ABI ("top-lazy2", lab build) and k_trace_refill (RODENT_HIP_LAZY_MISS), parity and A/B
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
OUT=gpurun_out/r06
RODENT_HIP_LAB=1 timeout 900 python scripts/defer_experiment.py --only 'top-lazy2' --big --steps 40 2>&1 | grep -v amdgpu.ids | tee $OUT/lazy_miss_traversal.txt
( for rep in 1 2; do for z in 0 1; do echo "== RODENT_HIP_LAZY_MISS=$z atrium 3840x2160 x 64 spp"; RODENT_HIP_LAZY_MISS=$z timeout 600 python scripts/frame_rate.py --spp 64; done; done
  for z in 0 1; do echo "== RODENT_HIP_LAZY_MISS=$z gallery 16 spp"; RODENT_HIP_LAZY_MISS=$z timeout 600 python scripts/frame_rate.py --scene gallery --spp 16; done
  for z in 0 1; do echo "== RODENT_HIP_LAZY_MISS=$z cornell streaming"; RODENT_HIP_LAZY_MISS=$z timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 --mapping streaming; done ) 2>&1 | grep -v "amdgpu.ids\|Missing material" | tee $OUT/lazy_miss_render.txt
RODENT_HIP_LAZY_MISS=1 timeout 1200 python -m pytest tests/test_gpu_render.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -4
