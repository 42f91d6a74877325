#!/bin/bash
# renderer checks: parity tests of the renderer, then the frame rates DESIGN.md quotes (cfg4 sorted / unsorted / megakernel, atrium 1080p)
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py -m gpu -x -q 2>&1 | tail -4
C="--scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8"
for args in "$C" "$A"; do echo "RODENT_HIP_LDS_IMAGE=0 rodent $args"; RODENT_HIP_LDS_IMAGE=0 timeout 300 rodent_amd/bin/rodent $args 2>&1 | tail -1; done | tee gpurun_out/r02/render_rates_no_lds_image.txt
for args in "$C" "$C --no-sort" "$C --target amdgpu-megakernel" "$A" "$A --no-sort" "$A --target amdgpu-megakernel"; do
  echo "rodent $args"; timeout 300 rodent_amd/bin/rodent $args 2>&1 | tail -1
done | tee gpurun_out/r02/render_rates.txt
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profiles/r02b_render -o rodent -- rodent_amd/bin/rodent $C > gpurun_out/profiles/r02b_render.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/profiles/r02b_render/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f} %")
PY
