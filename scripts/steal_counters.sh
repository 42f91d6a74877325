#!/bin/bash
# usage (GPU box, lab build): scripts/steal_counters.sh <tag> -- issue counters of the lab kernel k_bvh2_top_steal against whole chunks on both benchmark sets
TAG=${1:-r04}; OUT=gpurun_out/$TAG/steal_counters; mkdir -p $OUT; export TMPDIR=/tmp RODENT_HIP_LAB=1
for v in top-chunks steal; do
  V=$(python -c "from rodent_amd import abi; print(abi.variants(2).index('$v'))")
  for set in primary random; do
    timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/${v}_$set -o b -- \
      python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-render --only $set --variant $V > $OUT/${v}_$set.log 2>&1 || echo "pass $v $set failed"
  done
done
python - <<PY | tee gpurun_out/$TAG/steal_counters.txt
import csv, glob, re
from collections import defaultdict
print("kernel                 set      | VALU instr per launch  lane utilisation  waiting / wave cycles  wave cycles (quad-cycles) per launch")
for v in ("top-chunks", "steal"):
    for s in ("primary", "random"):
        f = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (v, s), recursive=True)
        agg = defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            m = re.search(r"(k_bvh2_top_\w+)<", r["Kernel_Name"])
            if m and "finish" not in m.group(1):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        c = {k: sum(x) / len(x) for k, x in agg.items()}
        print(f"{v:22s} {s:8s} | {c['SQ_INSTS_VALU'] / 1e6:18.1f} M {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):17.3f} {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:22.3f} {c['SQ_WAVE_CYCLES'] / 1e6:14.1f} M")
PY
