#!/usr/bin/env python
"""BVH2 / BVH4 / BVH8 default mappings side by side on the benchmark scene: 1 Mi camera rays (closest hit, any hit), 1 Mi ao rays (any hit),
1 Mi random segments (closest hit, any hit).
ms per launch from one event pair around 30 launches, best of 3.  usage: python scripts/width_compare.py [scene]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
path = scenes.scene_bvh(scene)
eye, d, up, fov = scenes.CAMERAS[scene.split("/")[0]]
st = torch.cuda.current_stream()
n2, t2 = F.read_bvh(path, F.BVH2_TRI1)
lo, hi = raygen.scene_bounds2(n2)
prim = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX)
bvh2 = abi.DeviceBvh(2, n2, t2, 0)
hits = abi.traverse(bvh2, prim)
sets = {"camera": prim, "ao": raygen.shadow_rays(scenes.LIGHTS[scene.split("/")[0]], prim, hits["t"], 0.0, 0.999),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, scenes.RANDOM_TMAX)}
bvhs = {2: bvh2, 4: abi.DeviceBvh.load(path, 4, 0), 8: abi.DeviceBvh.load(path, 8, 0)}


def timed(bvh, rd, hd, n, any_hit):
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, any_hit, 0, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            abi.traverse_async(bvh, rd, hd, n, any_hit, 0, st)
        e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    return best


print(f"== {scene}: ms per launch of 1 Mi rays, default mapping of each layout        BVH2      BVH4      BVH8")
for name, rays in sets.items():
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for any_hit in ((False, True) if name != "ao" else (True,)):
        print(f"   {name:8s} {'any hit    ' if any_hit else 'closest hit'}                                              "
            + "  ".join(f"{timed(bvhs[w], rd, hd, n, any_hit):8.4f}" for w in (2, 4, 8)), flush=True)
abi.check_errors(0)
