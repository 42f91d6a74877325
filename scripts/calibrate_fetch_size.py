#!/usr/bin/env python
"""Digest of scripts/calibrate_fetch_size.sh: per vmem_peak dispatch (its second, timed launch of every pattern) the known 64-byte node
fetches against FETCH_SIZE (KiB units of 1024 B) and the TCC counters."""
import csv, sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])


def per_dispatch(sub):
    f = next(iter(sorted((out / sub).rglob("*counter_collection.csv"))), None)
    rows = defaultdict(dict)
    if f:
        for r in csv.DictReader(open(f)):
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"],
                0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


fetch, tcc = per_dispatch("fetch"), per_dispatch("tcc")
patterns = ["coherent (4 lanes share a node), 2 MiB set", "coherent (16 lanes share a node), 2 MiB set", "scattered, 16 KiB set (L1)",
    "scattered, 2 MiB set (L2)", "scattered, 64 MiB set (MALL)"]
combos = [("lane-per-node", 64), ("lane-per-node", 32), ("lane-per-node", 16), ("4-lanes-per-node", 64)]
waves, steps = 256 * 32, 256
print(f"{len(fetch)} / {len(tcc)} dispatches (expected {2 * len(patterns) * len(combos)}: every pattern is launched twice, the second "
    f"launch is listed)")
print(f"{'pattern':46s} {'form':18s} {'lanes':>5s} | {'nodes fetched':>14s} {'x 64 B (MB)':>12s} | {'FETCH_SIZE (MB)':>15s} "
    f"{'bytes / FETCH_SIZE':>18s} | {'TCC hit rate':>12s} {'TCC_MISS x 128 B (MB)':>21s} {'EA0_RDREQ x 64 B (MB)':>21s}")
k = 0
for p in patterns:
    for form, lanes in combos:
        d = 2 * k + 1
        k += 1
        if d >= len(fetch):
            continue
        nodes = waves * steps * lanes
        fs = fetch[d].get("FETCH_SIZE", 0.0) * 1024
        t = tcc[d] if d < len(tcc) else {}
        hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
        print(f"{p:46s} {form:18s} {lanes:5d} | {nodes:14d} {nodes * 64 / 1e6:12.1f} | {fs / 1e6:15.1f} {nodes * 64 / max(fs, 1.0):18.2f} | "
            f"{hit / max(hit + miss, 1.0):12.3f} {miss * 128 / 1e6:21.1f} {t.get('TCC_EA0_RDREQ_sum', 0.0) * 64 / 1e6:21.1f}")
