mkdir -p gpurun_out/prof; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r1 -- $B > gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/prof/pmcA -o a -- $B > gpurun_out/prof/pmcA.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d gpurun_out/prof/pmcB -o b -- $B > gpurun_out/prof/pmcB.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof/pmcC -o c -- $B > gpurun_out/prof/pmcC.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmcD -o d -- $B > gpurun_out/prof/pmcD.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmcE -o e -- $B > gpurun_out/prof/pmcE.log 2>&1
find gpurun_out/prof -name "*.db" -size +20M -delete
ls -R gpurun_out/prof | head -50
