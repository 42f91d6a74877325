#!/bin/bash
# round 5, call AJ: the gathered per-triangle shading record (SceneDev::tri_shade) against the path through indices -> normals (RODENT_HIP_TRI_SHADE=0), one build
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -3
( for rep in 1 2 3; do
    echo "== indices -> normals rep $rep"; RODENT_HIP_TRI_SHADE=0 timeout 600 python scripts/frame_rate.py --spp 64
    echo "== tri_shade rep $rep"; timeout 600 python scripts/frame_rate.py --spp 64
  done
  echo "== indices -> normals, gallery"; RODENT_HIP_TRI_SHADE=0 timeout 600 python scripts/frame_rate.py --scene gallery --spp 16
  echo "== tri_shade, gallery"; timeout 600 python scripts/frame_rate.py --scene gallery --spp 16
  echo "== indices -> normals, cornell (megakernel)"; RODENT_HIP_TRI_SHADE=0 timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4
  echo "== tri_shade, cornell (megakernel)"; timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/tri_shade_ab.txt; cat gpurun_out/r05/tri_shade_ab.txt
