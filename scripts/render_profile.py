#!/usr/bin/env python
"""Digest of scripts/render_profile.sh: per renderer kernel and mapping of one BASELINE render configuration -- calls per frame,
average duration (rocprofv3 --kernel-trace --stats), HBM bytes per call (FETCH_SIZE and WRITE_SIZE from separate --pmc passes,
FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950) -- as text and as profiles/<tag>_render_profile_<cfg>.json,
stamped with the hash of the renderer's sources (rodent_amd/provenance.py): bench.py quotes it only while that hash holds.
usage: python scripts/render_profile.py <out dir> <tag> <cfg4|cfg5> <frames> "<command>" """
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import provenance

out, tag, cfg, frames, command = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
clean = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
result = {"_meta": dict(provenance.stamp("render"), command=command, frames=frames)}
for mapping in ("streaming", "megakernel"):
    f = sorted(glob.glob(f"{out}/{tag}_rp_{cfg}_{mapping}_trace/**/*kernel_stats.csv", recursive=True))
    if not f:
        continue
    dur = {clean(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f[0]))}
    c = defaultdict(dict)
    for name in ("fetch", "write"):
        for g in glob.glob(f"{out}/{tag}_rp_{cfg}_{mapping}_{name}/**/*counter_collection.csv", recursive=True):
            agg = defaultdict(list)
            for r in csv.DictReader(open(g)):
                agg[(clean(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, cn), v in agg.items():
                c[k][cn] = sum(v) / len(v)
    print(f"== {cfg} {mapping}: {command}")
    print(f"{'kernel':34s} {'calls/frame':>11s} {'avg us':>9s} {'ms/frame':>9s} {'FETCH MB':>9s} {'WRITE MB':>9s} "
        f"{'HBM TB/s (2 x FETCH + WRITE)':>30s}")
    res = {}
    for k, (calls, us) in sorted(dur.items(), key=lambda x: -x[1][0] * x[1][1]):
        if not k.startswith("k_"):
            continue
        row = {"calls_per_frame": round(calls / frames, 2), "avg_us": us}
        line = f"{k[:34]:34s} {calls / frames:11.1f} {us:9.1f} {calls / frames * us / 1e3:9.3f}"
        if "FETCH_SIZE" in c.get(k, {}):
            fe, wr = c[k]["FETCH_SIZE"] * 1024 / 1e6, c[k].get("WRITE_SIZE", 0) * 1024 / 1e6
            row.update(fetch_MB=fe, write_MB=wr, hbm_TBps_fetch_x2=(2 * fe + wr) / us)
            line += f" {fe:9.1f} {wr:9.1f} {(2 * fe + wr) / us:30.3f}"
        print(line)
        res[k] = row
    result[mapping] = res
json.dump(result, open(f"{out}/{tag}_render_profile_{cfg}.json", "w"), indent=1)
