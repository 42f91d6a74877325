#!/bin/bash
# usage: scripts/render_profile.sh <tag> <cfg4|cfg5>   (on the GPU box through gpurun)
# Per-kernel durations (--kernel-trace --stats) and HBM traffic (FETCH_SIZE / WRITE_SIZE, each in its own --pmc pass, no trace
# domain in those runs) of the renderer CLI on one BASELINE render configuration, streaming and megakernel mapping.
TAG=${1:-r03}; CFG=${2:-cfg4}; OUT=gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$CFG" = cfg4 ]; then FR=2; C="rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench $FR --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
else python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"; FR=1; C="rodent_amd/bin/rodent --scene data/atrium.obj --bench $FR --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 3840 --height 2160 --spp 256 --max-path-len 8"; fi
for M in streaming megakernel; do
  run() { name=$1; shift; timeout -k 5 600 rocprofv3 "$@" --output-format csv -d $OUT/${TAG}_rp_${CFG}_${M}_$name -o rodent -- $C --target amdgpu-$M > $OUT/${TAG}_rp_${CFG}_${M}_$name.log 2>&1 || echo "pass $M $name failed"; }
  run trace --kernel-trace --stats
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
done
python scripts/render_profile.py $OUT $TAG $CFG $FR "$C --target amdgpu-<mapping>" | tee $OUT/${TAG}_render_profile_${CFG}.txt
