#!/usr/bin/env python
"""One ray set, one size, the default mapping: a few launches with HIP-event times -- the command scripts/big_launch_counters.sh puts under
rocprofv3 --pmc to get the issue counters of the 16 Mi-ray launch (the valu_issue block at the size where the kernel is
not a tail). usage: python scripts/big_launch_counters.py [--side 4096] [--random N]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--side", type=int, default=4096)
ap.add_argument("--random", type=int, default=0)
ap.add_argument("--launches", type=int, default=6)
a = ap.parse_args()
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
if a.random:
    n4, _ = F.read_bvh(path, F.BVH4_TRI4)
    lo, hi = raygen.scene_bounds(n4)
    rays = raygen.random_rays(lo, hi, a.random, 43, 0.0, 1.0)
else:
    eye, d, up, fov = scenes.CAMERAS["atrium"]
    rays = raygen.primary_rays(eye, d, up, fov, a.side, a.side, 0.0, 5000.0)
n = len(rays)
rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream()
ms = []
for k in range(a.launches):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(st); abi.traverse_async(bvh, rd, hd, n, False, 0, st); e.record(st)
    torch.cuda.synchronize(); ms.append(s.elapsed_time(e))
print(f"{n} rays per launch, kernel {abi.kernel_name(2, 0)}: ms per launch {[round(x, 4) for x in ms]}")
