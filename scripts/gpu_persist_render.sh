#!/bin/bash
# persistent stream traversal kernels in the renderer: parity (atrium tests under the switch), frame rates with / without
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
RODENT_HIP_TRACE_PERSISTENT=1 timeout 900 python -m pytest tests/test_gpu_atrium.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -3
C="--scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8"
for P in 0 1; do for args in "$C" "$A" "$A --no-sort"; do echo "RODENT_HIP_TRACE_PERSISTENT=$P rodent $args"; RODENT_HIP_TRACE_PERSISTENT=$P timeout 300 rodent_amd/bin/rodent $args 2>&1 | tail -1; done; done | tee gpurun_out/r02/render_rates_persistent.txt
