#!/usr/bin/env python
"""Predicts the strong-scaling curve of the traversal benchmark on ONE GPU: the 1 Mi-ray set is divided among
2 / 4 / 8 ranks as bench.py / bench_traversal -ngpu do it (parallel.ray_range: contiguous ranges), every rank's share is traced alone
and timed (HIP events, default mapping); an iteration takes as long as the slowest share, so predicted Mrays/s = rays / max.
For comparison: shares of interleaved 2048-ray groups (rank r takes groups r, r + N, ...), which balance the cost but not the floor.
usage: python scripts/range_costs.py [--steps 30]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, parallel, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--variants", default="top",
    help="comma-separated mapping names (lab build: e.g. top,fast-pf0,fast-pf48): one table per mapping")
ap.add_argument("--worlds", default="2,4,8")
a = ap.parse_args()
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}


VARIANT = 0


def timed(rays):
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, False, VARIANT, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, VARIANT, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


for vname, name, rays in [(v, k, r) for v in a.variants.split(",") for k, r in sets.items()]:
    VARIANT = abi.variants(2).index(vname)
    n = len(rays)
    one = timed(rays)
    if vname != "top":
        print(f"== mapping {vname}")
    print(f"{name}: {n} rays on one GPU {one:.4f} ms = {n / one / 1e3:.0f} Mrays/s")
    print(f"  {'GPUs':>4s} {'partition':>24s} | per-rank ms" + " " * 52
        + "| mean / max      | predicted Mrays/s (Hit1 gather outside the timed region)")
    for world in [int(x) for x in a.worlds.split(",")]:
        for kind in ("contiguous ranges", "interleaved 2048-ray groups"):
            ms = []
            for r in range(world):
                if kind.startswith("contiguous"):
                    lo_, hi_ = parallel.ray_range(n, r, world)
                    share = rays[lo_:hi_]
                else:
                    g = np.arange(r, (n + 2047) // 2048, world)
                    idx = (g[:, None] * 2048 + np.arange(2048)[None, :]).ravel()
                    share = rays[idx[idx < n]]
                ms.append(timed(np.ascontiguousarray(share)))
            print(f"  {world:4d} {kind:>24s} | " + " ".join(f"{x:7.4f}"
                for x in ms).ljust(63) + f"| {np.mean(ms):.4f} / {max(ms):.4f} | {n / max(ms) / 1e3:8.0f}  ({one / max(ms):.2f} x one GPU)", flush=True)
