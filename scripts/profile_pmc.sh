#!/bin/bash
# usage: scripts/profile_pmc.sh <tag> [variant]   (on the GPU box through gpurun)
# Memory-pipeline and issue counters of the traversal kernel on the primary and the random pass, one small group per
# rocprofv3 --pmc pass (no tracing in the same run): which unit is busy while the kernel runs.
TAG=${1:-r02}; VARIANT=${2:--1}; OUT=gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
for SET in primary random; do
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only $SET --variant $VARIANT"
run() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_${SET}_$name -o bench -- $B > $OUT/${TAG}_pmc_${SET}_$name.log 2>&1 || echo "pass $SET $name failed"; }
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 TA_BUSY_avr TA_BUSY_max
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT
run sq4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
done
python scripts/pmc_digest.py $OUT ${TAG}_pmc > $OUT/${TAG}_pmc_digest.txt 2>&1
tail -150 $OUT/${TAG}_pmc_digest.txt
