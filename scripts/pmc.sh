#!/bin/bash
# usage: scripts/pmc.sh <tag> <bvh-width> <variant> <primary|random>
# Collects rocprofv3 kernel trace + PMC passes (one counter group per run) for one kernel variant
# on one ray set, and prints a compact summary.  Outputs under gpurun_out/pmc/<tag>/.
TAG=$1; W=$2; V=$3; ONLY=$4
OUT=gpurun_out/pmc/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --bvh-width $W --variant $V --only $ONLY"
run() { name=$1; shift; timeout -k 5 90 rocprofv3 "$@" -d $OUT/$name -o r -- $B > $OUT/$name.log 2>&1 || echo "pass $name failed/timeout"; }
run trace --kernel-trace --stats
run p1 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_LEVEL_WAVES
run p2 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
# (TA_* counters abort rocprofv3 on this image and hang the run: not collected)
run p4 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
run p5 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
run p6 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run p7 --pmc FETCH_SIZE
run p8 --pmc WRITE_SIZE
python scripts/rocpd_summary.py $OUT/*/r_results.db > $OUT/summary.txt 2>&1
grep -E "k_bvh" $OUT/summary.txt | cut -c1-40,52-200 | sort -u
