export TMPDIR=/tmp; mkdir -p gpurun_out/r03
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
C="--scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4 --target amdgpu-megakernel"
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8 --target amdgpu-megakernel"
for J in 0 1; do echo "RODENT_HIP_MEGA_CURSOR=$J: cfg4 $(RODENT_HIP_MEGA_CURSOR=$J rodent_amd/bin/rodent $C | tail -1)   atrium1080p16 $(RODENT_HIP_MEGA_CURSOR=$J rodent_amd/bin/rodent $A | tail -1)"; done
RODENT_HIP_MEGA_CURSOR=1 timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py -m gpu -x -q -k "mega" 2>&1 | tail -3
