#!/bin/bash
# usage (GPU box): scripts/big_launch_counters.sh <tag>   -- SQ issue counters of the default kernel at 16 Mi camera rays and 8 Mi random segments per launch
TAG=${1:-r04}; OUT=gpurun_out/$TAG/big_counters; mkdir -p $OUT; export TMPDIR=/tmp
for set in "primary:--side 4096" "random:--random 8388608"; do
  name=${set%%:*}; args=${set#*:}
  timeout -k 5 300 python scripts/big_launch_counters.py $args > $OUT/${name}_plain.log 2>&1
  timeout -k 5 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/$name -o b -- python scripts/big_launch_counters.py $args > $OUT/${name}_pmc.log 2>&1 || echo "pass $name failed"
done
python - <<PY | tee gpurun_out/$TAG/big_launch_counters.txt
import csv, glob, re
from collections import defaultdict
for name in ("primary", "random"):
    plain = open("$OUT/%s_plain.log" % name).read().strip().splitlines()[-1]
    ms = [float(x) for x in re.findall(r"[0-9.]+", plain.split("[")[-1])]
    med = sorted(ms[1:])[len(ms[1:]) // 2]
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_bvh2_top" in r["Kernel_Name"] and "finish" not in r["Kernel_Name"]:
            agg[(re.search(r"(k_bvh2_top_\w+)<", r["Kernel_Name"]).group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(plain)
    kernels = sorted({k for k, _ in agg})
    for k in kernels:
        c = {cn: sum(v) / len(v) for (kk, cn), v in agg.items() if kk == k}
        n = len(agg[(k, "SQ_INSTS_VALU")])
        rate = c["SQ_INSTS_VALU"] / (med * 1e3) / 1024
        print(f"  {k} ({n} launches): SQ_INSTS_VALU {c['SQ_INSTS_VALU'] / 1e6:.1f} M per launch -> {rate:.0f} wave-instructions per us per SIMD at the unprofiled median {med:.4f} ms "
              f"= {rate / 771:.2f} of the highest measured mix (771), {rate / 645:.2f} of the loop-mix ceiling (645), {rate / 1162:.2f} of the guide's 2-cycle rate; "
              f"lane utilisation {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}; waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} of the wave cycles")
PY
