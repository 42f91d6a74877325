#!/bin/bash
# extra evidence for DESIGN 3.1-3.3: launch-size scaling of the default kernel, SQ counters of the BVH4 / BVH8 kernels and of the sorted mapping
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
timeout 300 python scripts/batch_scaling.py > gpurun_out/r02/batch_scaling.txt 2>&1; cat gpurun_out/r02/batch_scaling.txt
for W in 4 8; do
  timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/profiles/r02_pmcw_primary_bvh$W -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only primary --bvh-width $W > gpurun_out/profiles/r02_pmcw_primary_bvh$W.log 2>&1
done
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profiles/r02_trace_sorted -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only random > gpurun_out/profiles/r02_trace_sorted.log 2>&1
python scripts/pmc_digest.py gpurun_out/profiles r02_pmcw k_wide | cut -c1-140
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/profiles/r02_trace_sorted/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
