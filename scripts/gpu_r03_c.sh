#!/bin/bash
# fused compaction: look-back scan (1) against one atomic per block (2) against the separate pass (0); kernels alone (OVERLAP=0) and the frame rates
mkdir -p gpurun_out/r03 gpurun_out/profiles; export TMPDIR=/tmp
C="rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench 3 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4 --target amdgpu-streaming"
for F in 0 1 2; do
echo "FUSED_COMPACT=$F: overlap on: $(RODENT_HIP_FUSED_COMPACT=$F $C | tail -1)   overlap off: $(RODENT_HIP_OVERLAP=0 RODENT_HIP_FUSED_COMPACT=$F $C | tail -1)"
RODENT_HIP_OVERLAP=0 RODENT_HIP_FUSED_COMPACT=$F timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profiles/r03c_fc$F -o rodent -- $C > gpurun_out/profiles/r03c_fc$F.log 2>&1
python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/profiles/r03c_fc$F/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.5: print(f"   {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f} %")
PY
done 2>&1 | tee gpurun_out/r03/fused_compaction_modes.txt
