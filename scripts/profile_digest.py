#!/usr/bin/env python
"""Condenses the rocprofv3 CSV outputs of scripts/profile_round.sh into one text digest + a JSON with
the per-launch HBM traffic of the dominant traversal kernel (read by bench.py for roofline.traffic)."""
import csv, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import provenance

out, tag = Path(sys.argv[1]), sys.argv[2]


def kernel_stats(path, title):
    f = next(iter(sorted(path.rglob("*kernel_stats.csv"))), None)
    if not f:
        print(f"[{title}] no kernel_stats.csv"); return {}
    print(f"== {title}: rocprofv3 --kernel-trace --stats ({f.name})")
    print(f"{'kernel':78s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    res = {}
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{name[:78]:78s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} "
              f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}")
        res[name] = float(r["AverageNs"])
    return res


def counters(path):
    f = next(iter(sorted(path.rglob("*counter_collection.csv"))), None)
    agg = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""),
                r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


kernel_stats(out / f"{tag}_trace", "bench.py (primary + random passes)")
kernel_stats(out / f"{tag}_trace_primary", "bench.py --only primary (the pass `value` and `roofline` are quoted on)")
kernel_stats(out / f"{tag}_render", "rodent cfg4 (Cornell 1920x1080, 64 spp, max path length 4)")
kernel_stats(out / f"{tag}_render_mega", "rodent cfg4, --target amdgpu-megakernel")
traffic = {}
for sub, title in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("tcc", "TCC"), ("sq", "SQ (primary)"), ("sqr", "SQ (random)")):
    c = counters(out / f"{tag}_{sub}")
    print(f"== PMC pass {title}: mean per dispatch")
    for (k, name), (mean, n) in sorted(c.items()):
        if "k_bvh2" in k:
            print(f"   {k[:60]:60s} {name:24s} n={n:3d} {mean:16.1f}")
            traffic.setdefault(k.split("(")[0], {})[("random_" + name) if sub == "sqr" else name] = mean
for k, t in traffic.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        # FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B; the guide's gfx950 correction (x2 on wide coalesced
        # reads) is reported separately because node fetches here are 16-B scattered loads (uncalibrated).
        t["hbm_bytes_raw"] = (t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024
        t["hbm_bytes_fetch_x2"] = (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024
        print(f"== traffic per launch [{k}]: raw {(t['hbm_bytes_raw']) / 1e6:.1f} MB, with x2 FETCH correction "
            f"{t['hbm_bytes_fetch_x2'] / 1e6:.1f} MB")
traffic["_meta"] = provenance.stamp("traversal")      # bench.py quotes the traffic only while this hash holds
json.dump(traffic, open(out / f"{tag}_traffic.json", "w"), indent=1)
