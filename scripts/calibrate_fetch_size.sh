#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE on the traversal kernels' access pattern: scripts/ubench/vmem_peak fetches a KNOWN number of
# 64-byte nodes, one per lane, scattered over working sets that fit L1 / L2 / only the Infinity Cache; FETCH_SIZE and the TCC hit / miss / request
# counters are collected per dispatch in separate --pmc passes (gpurun refuses nothing here: no trace domains).  The guide calibrates the x2 only for
# wide streaming reads.  usage (GPU box): scripts/calibrate_fetch_size.sh <tag>
TAG=${1:-r04}; OUT=gpurun_out/$TAG/fetch_calibration; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o v -- rodent_amd/bin/vmem_peak > $OUT/fetch.log 2>&1 || echo "FETCH_SIZE pass failed"
timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/tcc -o v -- rodent_amd/bin/vmem_peak > $OUT/tcc.log 2>&1 || echo "TCC pass failed"
python scripts/calibrate_fetch_size.py $OUT | tee gpurun_out/$TAG/fetch_size_calibration.txt
