#!/bin/bash
# round 3, call B: renderer with the compaction fused into the shader and the per-scene mapping choice -- parity, A/B frame rates,
# the mapping cross-over, then the new bench.py end to end
mkdir -p gpurun_out/r03; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_atrium.py -m gpu -x -q 2>&1 | tail -6
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
C="--scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8"
for env in "RODENT_HIP_FUSED_COMPACT=0" "RODENT_HIP_FUSED_COMPACT=1"; do
  for args in "$C --target amdgpu-streaming" "$C --target amdgpu-streaming --no-sort" "$A --target amdgpu-streaming" "$A --target amdgpu-streaming --no-sort" "$C" "$A"; do
    echo "$env rodent $args"; env $env timeout 300 rodent_amd/bin/rodent $args 2>&1 | tail -1
  done
done | tee gpurun_out/r03/render_rates_fused_compaction.txt
timeout 900 python scripts/mapping_sweep.py 2>&1 | tee gpurun_out/r03/mapping_sweep.txt
timeout 900 python bench.py > gpurun_out/r03/bench_b.json 2> gpurun_out/r03/bench_b.err; tail -3 gpurun_out/r03/bench_b.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_b.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "scaling")}, d["roofline"]["bound"], d["roofline"]["frac"], d["extra"]["all_rays_bit_exact_vs_oracle"])
print(json.dumps(d["extra"]["render"], indent=1)[:6000])
PY
