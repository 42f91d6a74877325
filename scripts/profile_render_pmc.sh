#!/bin/bash
# usage: scripts/profile_render_pmc.sh <tag>: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and L1 / L2 counters of the renderer's
# kernels on cfg4 (Cornell 1920x1080, 64 spp, path length 4), plus the per-kernel durations of the same command.
TAG=${1:-r02}; OUT=gpurun_out/profiles; mkdir -p $OUT; export TMPDIR=/tmp
C="rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench 2 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
run() { name=$1; shift; timeout -k 5 200 rocprofv3 "$@" --output-format csv -d $OUT/${TAG}_rpmc_$name -o rodent -- $C > $OUT/${TAG}_rpmc_$name.log 2>&1 || echo "pass $name failed"; }
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - <<PY
import csv, glob, json
from collections import defaultdict
out, tag = "$OUT", "$TAG"
dur = {}
f = sorted(glob.glob(f"{out}/{tag}_rpmc_trace/**/*kernel_stats.csv", recursive=True))
for r in csv.DictReader(open(f[0])):
    dur[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
c = defaultdict(dict)
for name in ("fetch", "write", "tcp", "tcc"):
    for f in glob.glob(f"{out}/{tag}_rpmc_{name}/**/*counter_collection.csv", recursive=True):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, cn), v in agg.items():
            c[k][cn] = sum(v) / len(v)
print(f"{'kernel':28s} {'calls':>6s} {'avg us':>9s} {'FETCH MB':>9s} {'WRITE MB':>9s} {'HBM TB/s (2xFETCH+WRITE)':>26s} {'L1 hit':>7s} {'L2 hit':>7s}")
res = {}
for k, (calls, us) in sorted(dur.items(), key=lambda x: -x[1][0] * x[1][1]):
    if k not in c or "FETCH_SIZE" not in c[k]:
        continue
    fe, wr = c[k]["FETCH_SIZE"] * 1024 / 1e6, c[k].get("WRITE_SIZE", 0) * 1024 / 1e6
    l1 = 1 - c[k].get("TCP_TCC_READ_REQ_sum", 0) / max(c[k].get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1)
    l2 = c[k].get("TCC_HIT_sum", 0) / max(c[k].get("TCC_HIT_sum", 0) + c[k].get("TCC_MISS_sum", 0), 1)
    tb = (2 * fe + wr) / us / 1e6 * 1e6 / 1e6
    print(f"{k[:28]:28s} {calls:6d} {us:9.1f} {fe:9.1f} {wr:9.1f} {(2 * fe + wr) / us:26.3f} {l1:7.3f} {l2:7.3f}")
    res[k] = {"calls": calls, "avg_us": us, "fetch_MB": fe, "write_MB": wr, "hbm_TBps_fetch_x2": (2 * fe + wr) / us, "l1_hit": l1, "l2_hit": l2}
json.dump(res, open(f"{out}/{tag}_render_traffic.json", "w"), indent=1)
PY
