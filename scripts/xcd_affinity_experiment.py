#!/usr/bin/env python
"""XCD-affine placement of incoherent rays.  Each XCD has its own 4 MB L2 and the kernels hand 2048-ray groups to the
stripes round-robin (group j -> stripe j % 64 -> XCD j % 8), so every L2 sees rays from everywhere in the 22 MB hierarchy.  The experiment
reorders the random segments ON THE HOST so that the rays whose origin (or midpoint) lies in octant x of the scene box land in groups of XCD
x -- eight bins and an affinity, no finer sort -- and times the unchanged kernels; controls: the same bins laid out contiguously (sorted,
but every XCD still sees every bin), and a random shuffle.  Hits are compared as sets (the rays are the same rays). usage: RODENT_HIP_LAB=1
python scripts/xcd_affinity_experiment.py [--steps 20]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
names = abi.variants(2)
variants = [("default (refill kernel after the first launch)", names.index("top"))] + ([("whole chunks", names.index("top-chunks"))]
    if "top-chunks" in names else [])


def timed(v, rays):
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    h = abi.from_device(hd, F.HIT1)
    return float(np.median([s.elapsed_time(e) for s, e in ev])), h


def octant(points):
    c = 0.5 * (lo + hi)
    return ((points[:, 0] > c[0]).astype(np.int64) | ((points[:, 1] > c[1]).astype(np.int64) << 1) | ((points[:,
        2] > c[2]).astype(np.int64) << 2))


def xcd_layout(bins, n):
    """order[k] = ray index: the rays of bin x fill the 2048-ray groups j with j % 8 == x in order; what does not fit its own XCD's groups
    fills the holes"""
    groups = (n + 2047) // 2048
    slots = [np.concatenate([np.arange(j * 2048, min((j + 1) * 2048, n)) for j in range(x, groups, 8)] or [np.zeros(0, np.int64)])
        for x in range(8)]
    order = np.full(n, -1, np.int64)
    spill = []
    for x in range(8):
        mine = np.nonzero(bins == x)[0]
        k = min(len(mine), len(slots[x]))
        order[slots[x][:k]] = mine[:k]
        spill.append(mine[k:])
    spill = np.concatenate(spill)
    holes = np.nonzero(order < 0)[0]
    order[holes] = spill
    return order


for count in (1 << 20, 1 << 23):
    rays = raygen.random_rays(lo, hi, count, 42, 0.0, 1.0)
    org = np.asarray(rays["org"], np.float64); mid = org + 0.5 * np.asarray(rays["dir"], np.float64)
    layouts = {"as generated": np.arange(count),
               "origin octant -> XCD": xcd_layout(octant(org), count),
               "midpoint octant -> XCD": xcd_layout(octant(mid), count),
               "origin octant, bins contiguous (control)": np.argsort(octant(org), kind="stable"),
               "origin octant -> XCD + 1 (control: affinity to another XCD is as good)": xcd_layout((octant(org) + 1) % 8, count)}
    print(f"{count} random segments")
    base = {}
    for label, order in layouts.items():
        assert np.array_equal(np.sort(order), np.arange(count))
        r = np.ascontiguousarray(rays[order])
        row = []
        for vname, v in variants:
            ms, h = timed(v, r)
            back = np.empty_like(h); back[order] = h
            base.setdefault(vname, back)
            row.append(f"{vname}: {ms:.4f} ms = {count / ms / 1e3:6.0f} Mrays/s (hits "
                f"{'identical' if back.tobytes() == base[vname].tobytes() else 'DIFFER'})")
        print(f"   {label:72s} " + "   ".join(row), flush=True)
