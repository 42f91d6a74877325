#!/usr/bin/env python
"""A/B of one library build (RODENT_HIP_LIB): the default mapping on the atrium's 1 Mi primary / random rays, plain and under 7 padding
levels (tests/conftest.pad_bvh2_depth: 1.3 % / 0.7 % of the rays outgrow the LDS window).  One line per build."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import torch
from conftest import pad_bvh2_depth
from rodent_amd import abi, formats as F, raygen, scenes
path = scenes.scene_bvh("atrium")
nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
eye, d, up, fov = scenes.CAMERAS["atrium"]
lo, hi = raygen.scene_bounds(F.read_bvh(path, F.BVH4_TRI4)[0])
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
bvhs = {"plain": abi.DeviceBvh(2, nodes, tris, 0), "deep": abi.DeviceBvh(2, pad_bvh2_depth(nodes, 7), tris, 0)}
st = torch.cuda.current_stream()
out = []
for kind, r in sets.items():
    n = len(r); rd = abi.to_device(r, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for name, bvh in bvhs.items():
        for _ in range(5): abi.traverse_async(bvh, rd, hd, n, False, 0, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(50): abi.traverse_async(bvh, rd, hd, n, False, 0, st)
        e1.record(st); torch.cuda.synchronize()
        out.append(f"{kind} {name} {e0.elapsed_time(e1) / 50:.4f}")
import os
print(f"{os.path.basename(os.environ.get('RODENT_HIP_LIB', 'librodent_hip.so')):22s} " + "  ".join(out), flush=True)
