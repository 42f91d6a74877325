#!/usr/bin/env python
"""Debug view of the adaptive launch order: per row (8 x 32 chunks) mean start time and mean wave life."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
rays = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0)
names = abi.variants(2)
abi.read_trace(arm_only=True)
for name in sys.argv[1:] or ["trace-fast"]:
    v = names.index(name)
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize(); abi.read_trace()
    abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize()
    tr = abi.read_trace()
    t0 = tr[:, 0].min(); start = (tr[:, 0] - t0) / 100.0; end = (tr[:, 1] - t0) / 100.0
    chunk = tr[:, 3].astype(np.int64); row = chunk // 256
    print(name, "span", end.max())
    order = []
    for r in range(64):
        m = row == r
        order.append((start[m].mean(), r, (end[m] - start[m]).mean(), end[m].max()))
    order.sort()
    print("  rows in order of mean start: " + " ".join(f"{r}({s:.0f}|{l:.0f}|{e:.0f})" for s, r, l, e in order))
