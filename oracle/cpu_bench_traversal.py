#!/usr/bin/env python
"""CPU side of bench_traversal (BASELINE.json config 0: "bench_traversal CPU single-ray ... plumbing, no GPU").

TEST INFRASTRUCTURE: the product CLI (rodent_amd/bin/bench_traversal) only contains the HIP variants; the CPU
variants of the reference (tools/bench_traversal/bench_traversal.cpp:44-122,307-327) are provided here on top of
the oracle / CPU baseline, with the same flags and the same stdout protocol (:294,381-391):
   -s / --single   single-ray kernel   (oracle B2 = mapping_cpu.impala:138-256; --bvh-width 4|8)
   default         hybrid ray8 x bvh8   (oracle/hybrid_baseline.cpp = mapping_cpu.impala:259-402)
   -any, --tmin, --tmax, --bench, --warmup, -o as in the reference.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import binding as O                 # noqa: E402
from rodent_amd import formats as F             # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(prog="bench_traversal (CPU)")
    ap.add_argument("-bvh", "--bvh-file", required=True)
    ap.add_argument("-ray", "--ray-file", required=True)
    ap.add_argument("--tmin", type=float, default=0.0)
    ap.add_argument("--tmax", type=float, default=1e9)
    ap.add_argument("--bench", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("-any", action="store_true")
    ap.add_argument("-s", "--single", action="store_true")
    ap.add_argument("-p", "--packet", action="store_true")
    ap.add_argument("--bvh-width", type=int, default=4)
    ap.add_argument("--ray-width", type=int, default=8)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("-o", "--output", default="")
    a = ap.parse_args(argv)
    if a.single and a.packet:
        print("Options '--packet' and '--single' are incompatible", file=sys.stderr); return 1
    if a.bvh_width not in (4, 8):
        print("Invalid BVH width", file=sys.stderr); return 1
    try:
        nodes, tris = F.read_bvh(a.bvh_file, F.BVH4_TRI4 if a.bvh_width == 4 else F.BVH8_TRI4)
    except (OSError, ValueError):
        print("Cannot load BVH file", file=sys.stderr); return 1
    try:
        rays = F.read_rays(a.ray_file, a.tmin, a.tmax)
    except (OSError, ValueError):
        print("Cannot load rays", file=sys.stderr); return 1
    if not a.single:
        rays = rays[: len(rays) // a.ray_width * a.ray_width]          # load_rays.h:74 drops the tail packet
    print(f"{len(rays)} ray(s) in the distribution file.")

    def run():
        t0 = time.perf_counter()
        if a.single or a.bvh_width == 4:
            hits, _ = O.traverse(a.bvh_width, nodes, tris, rays, any_hit=a.any)
        else:
            hits = O.cpu_baseline(nodes, tris, rays, any_hit=a.any, mode="hybrid", threads=a.threads)
        return (time.perf_counter() - t0) * 1e3, hits
    for _ in range(a.warmup):
        run()
    timings, hits = [], None
    for _ in range(max(a.bench, 1)):
        ms, hits = run()
        timings.append(ms)
    if a.output:
        F.write_fbuf(a.output, hits)
    timings.sort()
    total = sum(timings)
    print(f"{total:g}ms for {a.bench} iteration(s)")
    print(f"{len(rays) * a.bench / (1000.0 * total):g} Mrays/sec")
    print(f"# Average: {total / len(timings):g} ms")
    print(f"# Median: {timings[len(timings) // 2]:g} ms")
    print(f"# Min: {timings[0]:g} ms")
    print(f"{int((hits['tri_id'] >= 0).sum())} intersection(s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
