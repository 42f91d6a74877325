#!/bin/bash
# usage: scripts/profile_round.sh <round-tag>     (run on the GPU box through gpurun)
# Produces gpurun_out/profiles/<tag>_*: rocprofv3 kernel-trace stats of `python bench.py` and the
# HBM-traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) for the primary-ray pass.
TAG=${1:-r01}; OUT=gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o bench -- $B > $OUT/${TAG}_trace.log 2>&1
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o bench -- $B --only primary > $OUT/${TAG}_fetch.log 2>&1
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o bench -- $B --only primary > $OUT/${TAG}_write.log 2>&1
timeout -k 5 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/${TAG}_tcc -o bench -- $B --only primary > $OUT/${TAG}_tcc.log 2>&1
find $OUT -name "*.csv" | head -20
