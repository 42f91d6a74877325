#!/usr/bin/env python
"""What a launch costs before it traces anything: the Cornell box (16 nodes; a ray is a handful of steps) at ray counts from one chunk up,
through the one-chunk kernel ("fast"), the persistent LDS-image kernel forced (rodent_hip_top_min_rays(0): "top", "refill") and the wide
layouts' kernels.  ms per launch from one event pair around 50 back-to-back launches. usage: python scripts/fixed_costs.py"""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

st = torch.cuda.current_stream()
path = scenes.scene_bvh("cornell")
eye, d, up, fov = scenes.CAMERAS["cornell"]


def timed(bvh, rd, hd, n, v, steps=50):
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps):
            abi.traverse_async(bvh, rd, hd, n, False, v, st)
        e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    return best


abi.lib().rodent_hip_top_min_rays(0)
for width in (2, 4, 8):
    bvh = abi.DeviceBvh.load(path, width, 0)
    names = abi.variants(width)
    cols = [v for v in (os.environ.get("FIXED_COSTS_VARIANTS", "top,fast,single,refill").split(",")) if v in names]
    if not cols: continue
    print(f"== BVH{width}   rays  " + "  ".join(f"{c:>12s}" for c in cols))
    for n in (64, 4096, 65536, 262144, 393216, 1048576):
        side = int(np.sqrt(n)) if int(np.sqrt(n)) ** 2 == n else None
        w, h = (side, side) if side else (n // 256 if n >= 256 else n, 256 if n >= 256 else 1)
        rays = raygen.primary_rays(eye, d, up, fov, w, h, 0.0, 5000.0)
        assert len(rays) == n
        rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
        print(f"       {n:8d}  " + "  ".join(f"{timed(bvh, rd, hd, n, names.index(c)) * 1e3:10.1f}us" for c in cols), flush=True)
abi.check_errors(0)
