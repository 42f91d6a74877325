#!/usr/bin/env python
"""Times every kernel variant on the benchmark ray sets in one process (HIP events on the
launch stream) and checks that order-preserving variants stay bit-identical to variant 0.
usage: python scripts/sweep.py [--width 2|8] [--steps 20] [--variants 0,3,5]"""
import argparse, os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--variants", default=None)
ap.add_argument("--scene", default="atrium")
ap.add_argument("--any", action="store_true")
ap.add_argument("--reverse", action="store_true", help="experiment: trace the ray sets in reverse order")
a = ap.parse_args()

path = scenes.scene_bvh(a.scene)
bvh = abi.DeviceBvh.load(path, a.width, 0)
eye, d, up, fov = scenes.CAMERAS[a.scene]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
        "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
if a.reverse:
    sets = {k: np.ascontiguousarray(v[::-1]) for k, v in sets.items()}
names = abi.variants(a.width)
todo = [int(x) for x in a.variants.split(",")] if a.variants else range(len(names))
base = {}
print(f"{'variant':28s} " + " ".join(f"{k + ' ms':>12s} {k + ' Mr/s':>12s}" for k in sets) + "  identical-to-v0")
for v in todo:
    row, same = [], []
    for k, rays in sets.items():
        n = len(rays)
        rd = abi.to_device(rays, 0)
        hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
        st = torch.cuda.current_stream()
        for _ in range(3):
            abi.traverse_async(bvh, rd, hd, n, a.any, v, st)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for s, e in ev:
            s.record(st); abi.traverse_async(bvh, rd, hd, n, a.any, v, st); e.record(st)
        torch.cuda.synchronize()
        ms = float(np.median([s.elapsed_time(e) for s, e in ev]))
        row += [ms, n / ms / 1e3]
        h = abi.from_device(hd, F.HIT1)
        if k not in base:
            base[k] = h
        same.append(h.tobytes() == base[k].tobytes() if not a.any else bool(((h["tri_id"] >= 0) == (base[k]["tri_id"] >= 0)).all()))
    print(f"{v}:{names[v]:26s} " + " ".join(f"{x:12.4f}" for x in row) + f"  {same}", flush=True)
