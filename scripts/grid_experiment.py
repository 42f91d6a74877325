#!/usr/bin/env python
"""Camera rays as 8 x 8-pixel tiles instead of 64-pixel row segments per wave (RODENT_HIP_RAY_GRID=<image width>, experiment): kernel ms of
the default mapping on the primary set of a scene, and the hits' bytes against the run without it.  usage: RODENT_HIP_RAY_GRID=1024 python
scripts/grid_experiment.py [scene] [width]"""
import os, sys, hashlib
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nodes, tris = F.read_bvh(scenes.scene_bvh(scene), F.BVH2_TRI1)
bvh = abi.DeviceBvh(2, nodes, tris, 0)
eye, d, up, fov = scenes.CAMERAS[scene.split("/")[0]]
st = torch.cuda.current_stream()
for h in (w, w + 4):                                       # a height that is no multiple of 8: the last rows keep the linear mapping
    rays = raygen.primary_rays(eye, d, up, fov, w, h, 0.0, scenes.PRIMARY_TMAX)
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for _ in range(10):
        abi.traverse_async(bvh, rd, hd, n, False, 0, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            abi.traverse_async(bvh, rd, hd, n, False, 0, st)
        e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    got = abi.from_device(hd, F.HIT1)
    print(f"{scene} {w}x{h} grid={os.environ.get('RODENT_HIP_RAY_GRID', '0')}: {best:.4f} ms  {n / best / 1e3:.1f} Mrays/s  hits sha "
        f"{hashlib.sha1(got.tobytes()).hexdigest()[:12]}", flush=True)
abi.check_errors(0)
