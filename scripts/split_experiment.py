#!/usr/bin/env python
"""Intra-launch split: ONE 1 Mi-ray batch issued as K sub-launches on K streams, so that one part's fill
overlaps another's drain.  Parts are contiguous ranges or interleaved 32-chunk groups; every part goes through the default
mapping (which picks its kernel by the part's size) or is forced onto the persistent kernel (rodent_hip_top_min_rays(0)).
Time = first launch enqueued -> all parts done (HIP events: the parts' streams wait for a start event, an end stream waits for
all of them).  usage: python scripts/split_experiment.py"""
import sys
from pathlib import Path
import numpy as np
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
main = torch.cuda.current_stream()


def run(parts, streams, steps=30):
    """parts: list of (rays_dev, hits_dev, n)."""
    times = []
    for it in range(steps + 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for (rd, hd, n), st in zip(parts, streams):
            st.wait_event(e0)
            abi.traverse_async(bvh, rd, hd, n, False, 0, st)
            ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)
        e1.record(main)
        torch.cuda.synchronize()
        if it >= 4:
            times.append(e0.elapsed_time(e1))
    return float(np.median(times))


for name, rays in sets.items():
    n = len(rays)
    whole = [(abi.to_device(rays, 0), torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0"), n)]
    base = run(whole, [main])
    ref = abi.from_device(whole[0][1], F.HIT1)
    print(f"{name}: one launch {base:.4f} ms = {n / base / 1e3:.0f} Mrays/s")
    for k in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(k)]
        for layout in ("contiguous", "interleaved 32-chunk groups"):
            if layout == "contiguous":
                idx = [np.arange(n * j // k, n * (j + 1) // k) for j in range(k)]
            else:
                g = np.arange(n) // (32 * 64)
                idx = [np.nonzero(g % k == j)[0] for j in range(k)]
            parts = [(abi.to_device(rays[i], 0), torch.zeros(len(i) * 16, dtype=torch.uint8, device="cuda:0"), len(i)) for i in idx]
            for forced in (False, True):
                abi.lib().rodent_hip_top_min_rays(0 if forced else -1)
                ms = run(parts, streams)
                got = np.empty(n, F.HIT1)
                for i, (rd, hd, m) in zip(idx, parts):
                    got[i] = abi.from_device(hd, F.HIT1)[:m]
                print(f"   {k} parts, {layout:28s} "
                    f"{'persistent kernel forced' if forced else 'default mapping (one-chunk kernel at this size)':48s} {ms:.4f} ms "
                    f"({base / ms:.3f} x)  identical {got.tobytes() == ref.tobytes()}", flush=True)
    abi.lib().rodent_hip_top_min_rays(-1)
