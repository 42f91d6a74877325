#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_atrium.py tests/test_gpu_render.py tests/test_textures.py tests/test_services.py -x -q -m gpu 2>&1 | tail -3
bash scripts/gpu_r03_profiles_render.sh r03
