#!/usr/bin/env python
"""What ordering incoherent rays by the cell of their origin is worth to the CURRENT kernels (the follow-up of
scripts/xcd_affinity_experiment.py, whose control -- eight bins laid out contiguously -- gained 10 %): the random segments reordered ON THE
HOST into 2^3 / 4^3 / 8^3 / 16^3 cells of the scene box (cells in Morton order, rays inside a cell in their original order), traced by the
default mapping (the refill kernel, from the second launch on) and by whole chunks.  The time of the sort itself is NOT included: this
bounds what an in-launch counting sort may cost. usage: RODENT_HIP_LAB=1 python scripts/bin_order_experiment.py [--steps 20]"""
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
variants = [("default", names.index("top"))] + ([("whole chunks", names.index("top-chunks"))] if "top-chunks" in names else [])


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
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


def cell_key(points, bits):
    c = np.clip(((points - lo) / np.maximum(hi - lo, 1e-30) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(points), np.int64)
    for b in range(bits):
        for ax in range(3):
            key |= ((c[:, ax] >> b) & 1) << (3 * b + ax)
    return key


for count in (1 << 20, 1 << 23):
    rays = raygen.random_rays(lo, hi, count, 42, 0.0, 1.0)
    org = np.asarray(rays["org"], np.float64)
    print(f"{count} random segments")
    for label, order in [("as generated", np.arange(count))] + [(f"{1 << bits}^3 cells of the origin",
        np.argsort(cell_key(org, bits), kind="stable")) for bits in (1, 2, 3, 4)]:
        r = np.ascontiguousarray(rays[order])
        print(f"   {label:32s} " + "   ".join(f"{vn}: {timed(v, r):.4f} ms" for vn, v in variants), flush=True)
