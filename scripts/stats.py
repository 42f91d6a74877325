#!/usr/bin/env python
"""Runs the instrumented ("stats-*") kernel variants once per ray set and prints per-phase
iteration counts and SIMD lane utilisation."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
        "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
names = abi.variants(2)
abi.read_stats()
for v, name in enumerate(names):
    if not name.startswith("stats-"):
        continue
    for k, rays in sets.items():
        n = len(rays)
        rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
        abi.traverse_async(bvh, rd, hd, n, False, v)
        torch.cuda.synchronize()
        s = abi.read_stats()
        waves = n / 64
        print(f"{name:22s} {k:8s} node iters/wave {s[0]/waves:7.1f} lanes {100*s[1]/max(64*s[0],1):5.1f}% (per ray {s[1]/n:5.1f}) | "
              f"leaf iters/wave {s[2]/waves:6.1f} lanes {100*s[3]/max(64*s[2],1):5.1f}% (per ray {s[3]/n:4.2f}) | "
              f"refills {s[4]} lanes {s[5]} outer/wave {s[6]/waves:5.1f}")
