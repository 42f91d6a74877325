#!/bin/bash
# round 5, call AG: k_trace_refill with its streams as slabs (base + k x capacity) against the library of the last commit: renderer parity, then config 5's frame A/B
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -3
( for rep in 1 2 3; do
    echo "== exp_head rep $rep"; RODENT_HIP_LIB=rodent_amd/lib/exp_head.so timeout 600 python scripts/frame_rate.py --spp 64
    echo "== tree rep $rep"; timeout 600 python scripts/frame_rate.py --spp 64
  done
  echo "== exp_head 256 spp"; RODENT_HIP_LIB=rodent_amd/lib/exp_head.so timeout 600 python scripts/frame_rate.py --spp 256
  echo "== tree 256 spp"; timeout 600 python scripts/frame_rate.py --spp 256
  echo "== exp_head gallery"; RODENT_HIP_LIB=rodent_amd/lib/exp_head.so timeout 600 python scripts/frame_rate.py --scene gallery --spp 16
  echo "== tree gallery"; timeout 600 python scripts/frame_rate.py --scene gallery --spp 16 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/slab_ab.txt; cat gpurun_out/r05/slab_ab.txt
