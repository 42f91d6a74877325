#!/bin/bash
# round 6, call F: k_shade with the light record prefetched (exp_prefetch, RODENT_SHADE_HOIST_LIGHT=2) against the default build (no hoist)
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
OUT=gpurun_out/r06
( for rep in 1 2; do for lib in "" rodent_amd/lib/exp_prefetch.so; do echo "== RODENT_HIP_LIB=$lib atrium 3840x2160 x 64 spp"; RODENT_HIP_LIB=$lib timeout 600 python scripts/frame_rate.py --spp 64; done; done
  for lib in "" rodent_amd/lib/exp_prefetch.so; do echo "== RODENT_HIP_LIB=$lib gallery 16 spp"; RODENT_HIP_LIB=$lib timeout 600 python scripts/frame_rate.py --scene gallery --spp 16; done
  for lib in "" rodent_amd/lib/exp_prefetch.so; do echo "== RODENT_HIP_LIB=$lib cornell streaming"; RODENT_HIP_LIB=$lib timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 --mapping streaming; done ) 2>&1 | grep -v "amdgpu.ids\|Missing material" | tee $OUT/shade_prefetch_ab.txt
for lib in "" rodent_amd/lib/exp_prefetch.so; do
  tag=$( [ -z "$lib" ] && echo default || echo prefetch )
  RODENT_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shade_trace_$tag -o t -- python scripts/frame_rate.py --spp 16 --frames 2 > $OUT/shade_trace_$tag.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/shade_trace_$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_shade", "k_trace_refill", "k_generate")): print("$tag", r["Name"][:50], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "share", r["Percentage"])
PY
done
RODENT_HIP_LIB=rodent_amd/lib/exp_prefetch.so timeout 1200 python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -3
