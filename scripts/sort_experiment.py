#!/usr/bin/env python
"""Experiment: how much faster does the default kernel trace the RANDOM ray set when the rays arrive sorted for cache
locality (host-side sort, outside the timing)?  Keys: Morton code of the origin cell, origin + end point, midpoint,
origin cell + direction octant; at several grid resolutions."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
rays = raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


def morton(p, bits):
    q = np.clip(((p - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


def timed(r):
    n = len(r); rd = abi.to_device(r, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, 0, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, 0, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


org, end = rays["org"], rays["org"] + rays["dir"]
octant = ((rays["dir"][:, 0] < 0).astype(np.uint64) | ((rays["dir"][:, 1] < 0).astype(np.uint64) << 1) | ((rays["dir"][:,
    2] < 0).astype(np.uint64) << 2))
print(f"{'order':44s} {'ms':>8s} {'Mrays/s':>9s}")
print(f"{'file order':44s} {timed(rays):8.4f}")
for bits in (4,):
    keys = {f"morton(origin) {bits} bits/axis": morton(org, bits),
            f"morton(origin) {bits} b + morton(end) {min(bits, 4)} b": (morton(org, bits) << np.uint64(3 * min(bits, 4))) | morton(end,
                min(bits, 4)),
            f"morton(origin) {bits} b + octant": (morton(org, bits) << np.uint64(3)) | octant,
            f"morton(midpoint) {bits} bits/axis": morton(0.5 * (org + end), bits)}
    for name, k in keys.items():
        order = np.argsort(k, kind="stable")
        ms = timed(np.ascontiguousarray(rays[order]))
        print(f"{name:44s} {ms:8.4f} {len(rays) / ms / 1e3:9.1f}", flush=True)

# windowed sort: what a block-local sort in LDS (no global passes) would give
print("windowed sort by morton(origin) 10 bits + octant")
key = (morton(org, 10) << np.uint64(3)) | octant
for W in (1024, 4096, 16384, 65536, 262144):
    order = np.concatenate([s + np.argsort(key[s:s + W], kind="stable") for s in range(0, len(rays), W)])
    ms = timed(np.ascontiguousarray(rays[order]))
    print(f"window {W:7d} rays: {ms:8.4f} ms {len(rays) / ms / 1e3:9.1f} Mrays/s", flush=True)
