#!/usr/bin/env python
"""Mrays/s of the default traversal kernel as a function of the rays per launch (primary rays of the atrium camera at
w x w pixels): shows how much of a 1 Mi-ray launch is fill and drain."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, raygen, scenes
bvh = abi.DeviceBvh.load(scenes.scene_bvh("atrium"), 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
for w in (256, 512, 1024, 2048, 4096):
    rays = raygen.primary_rays(eye, d, up, fov, w, w, 0.0, 5000.0)
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, 0, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(st); abi.traverse_async(bvh, rd, hd, n, False, 0, st); b.record(st)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    print(f"{w:5d} x {w:<5d} = {n:9d} rays: {ms:8.4f} ms  {n / ms / 1e3:9.1f} Mrays/s")

# independent 1 Mi-ray batches in flight on several streams: the fill of one launch overlaps the drain of another
n = 1024 * 1024
rays = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0)
for streams in (1, 2, 3, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    rd = [abi.to_device(rays, 0) for _ in ss]; hd = [torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0") for _ in ss]
    for k, st_ in enumerate(ss):
        abi.traverse_async(bvh, rd[k], hd[k], n, False, 0, st_)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    launches = 0
    for _ in range(20):
        for k, st_ in enumerate(ss):
            abi.traverse_async(bvh, rd[k], hd[k], n, False, 0, st_); launches += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{streams} stream(s) x 1 Mi rays: {launches * n / dt / 1e6:9.1f} Mrays/s ({dt / launches * 1e3:.4f} ms per launch)")
