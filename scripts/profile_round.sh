#!/bin/bash
# usage: scripts/profile_round.sh <round-tag>     (run on the GPU box through gpurun)
# Produces gpurun_out/profiles/<tag>_*: rocprofv3 kernel-trace stats of `python bench.py`, the
# HBM-traffic counters (FETCH_SIZE / WRITE_SIZE / TCC, one group per pass) for the primary-ray pass,
# and kernel-trace stats of the renderer CLI (cfg4: Cornell 1920x1080, 64 spp, max path length 4).
TAG=${1:-r01}; OUT=gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout -k 5 150 rocprofv3 "$@" > $OUT/${TAG}_$name.log 2>&1 || echo "pass $name failed"; }
run trace --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o bench -- $B
run trace_primary --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_primary -o bench -- $B --only primary
run fetch --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o bench -- $B --only primary
run write --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o bench -- $B --only primary
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/${TAG}_tcc -o bench -- $B --only primary
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_sq -o bench -- $B --only primary
run sqr --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_sqr -o bench -- $B --only random
run render --kernel-trace --stats --output-format csv -d $OUT/${TAG}_render -o rodent -- rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4
run render_mega --kernel-trace --stats --output-format csv -d $OUT/${TAG}_render_mega -o rodent -- rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4 --target amdgpu-megakernel
python scripts/profile_digest.py $OUT $TAG > $OUT/${TAG}_digest.txt 2>&1
cat $OUT/${TAG}_digest.txt
