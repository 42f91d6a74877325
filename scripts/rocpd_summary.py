#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and PMC counter means.

usage: rocpd_summary.py results.db [more.db ...]   (prints a text summary)
"""
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(con):
    rows = con.execute("select name, (end - start) from kernels").fetchall()
    agg = defaultdict(list)
    for name, dur in rows:
        agg[name].append(dur)
    out = []
    for name, d in agg.items():
        d.sort()
        out.append((sum(d), name, len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[-1]))
    out.sort(reverse=True)
    return out


def pmc_stats(con):
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    if not cols:
        return []
    rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(list)
    for k, c, v in rows:
        agg[(k, c)].append(v)
    return sorted((k, c, len(v), sum(v) / len(v)) for (k, c), v in agg.items())


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        print(f"== {path}")
        ks = kernel_stats(con)
        if ks:
            print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'med_us':>10s} {'min_us':>10s} {'max_us':>10s}")
            for tot, name, n, avg, med, mn, mx in ks[:12]:
                print(f"{name[:70]:70s} {n:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {med / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f}")
        try:
            ps = pmc_stats(con)
        except sqlite3.Error as e:
            ps = []
        if ps:
            print(f"{'kernel':50s} {'counter':32s} {'n':>5s} {'mean/dispatch':>18s}")
            for k, c, n, mean in ps:
                if "bvh" in k or "k_" in k:
                    print(f"{k[:50]:50s} {c:32s} {n:5d} {mean:18.1f}")


if __name__ == "__main__":
    main()
