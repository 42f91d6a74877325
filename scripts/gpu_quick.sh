#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03; export TMPDIR=/tmp
rm -rf gpurun_out/r03/dense_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/dense_trace -o t -- python scripts/refill_sweep.py --idle 40 --scenes atrium --frames 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03/dense_trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
timeout 900 python scripts/refill_sweep.py --idle 40 --scenes atrium,atrium/8,cornell --frames 5 2>&1 | tail -3 | cut -c1-75
timeout 900 python scripts/refill_sweep.py --idle 40 --scenes atrium --size 3840x2160 --spp 32 --frames 3 2>&1 | tail -1 | cut -c1-75
