#!/usr/bin/env python
"""Where the default mapping should switch from the one-chunk kernel to the persistent LDS-image kernel: both at 64 Ki ... 1 Mi
primary rays of the atrium camera (rodent_hip_top_min_rays(0) forces the persistent kernel).
usage: python scripts/threshold_sweep.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, raygen, scenes

bvh = abi.DeviceBvh.load(scenes.scene_bvh("atrium"), 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
names = abi.variants(2)
st = torch.cuda.current_stream()


def timed(v, rd, hd, n, steps=30):
    for _ in range(4):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


abi.lib().rodent_hip_top_min_rays(0)
print(f"{'rays':>10s} {'fast ms':>9s} {'top ms':>9s}  top / fast")
for w, h in ((256, 256), (512, 256), (512, 384), (512, 512), (768, 512), (1024, 512), (1024, 768), (1024, 1024)):
    rays = raygen.primary_rays(eye, d, up, fov, w, h, 0.0, 5000.0)
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    f, t = timed(names.index("fast"), rd, hd, n), timed(names.index("top"), rd, hd, n)
    print(f"{n:10d} {f:9.4f} {t:9.4f}  {t / f:.3f}")
abi.lib().rodent_hip_top_min_rays(-1)
