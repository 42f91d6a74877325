#!/usr/bin/env python
"""Times the shipped kernel of every BVH layout (BVH2/Tri1, BVH4/Tri4, BVH8/Tri4) on the benchmark ray sets at 1 Mi rays
per launch and on 16 Mi primary rays per launch (HIP events on the launch stream, median of --steps), closest and any
hit, and checks every result against variant 0 of its layout.
usage: python scripts/sweep_widths.py [--steps 20] [--widths 2,4,8] [--all-variants] [--big] [--big-random]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--widths", default="2,4,8")
ap.add_argument("--all-variants", action="store_true")
ap.add_argument("--big", action="store_true", help="also 16 Mi primary rays per launch")
ap.add_argument("--mid", action="store_true", help="also 256 Ki, 2 Mi and 4 Mi primary rays per launch")
ap.add_argument("--big-random", action="store_true", help="also 8 Mi random segments per launch (what a renderer's bounce pass looks like)")
ap.add_argument("--only", default=None, help="comma-separated variant names")
ap.add_argument("--scene", default="atrium")
a = ap.parse_args()

path = scenes.scene_bvh(a.scene)
eye, d, up, fov = scenes.CAMERAS[a.scene]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
        "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
if a.big:
    sets["primary16Mi"] = raygen.primary_rays(eye, d, up, fov, 4096, 4096, 0.0, 5000.0)
if a.big_random:
    sets["random8Mi"] = raygen.random_rays(lo, hi, 1 << 23, 43, 0.0, 1.0)
if a.mid:
    sets["primary256Ki"] = raygen.primary_rays(eye, d, up, fov, 512, 512, 0.0, 5000.0)
    sets["primary2Mi"] = raygen.primary_rays(eye, d, up, fov, 2048, 1024, 0.0, 5000.0)
    sets["primary4Mi"] = raygen.primary_rays(eye, d, up, fov, 2048, 2048, 0.0, 5000.0)
dev = {k: abi.to_device(v, 0) for k, v in sets.items()}


def timed(bvh, k, any_hit, v):
    n = len(sets[k]); rd = dev[k]
    hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, any_hit, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, any_hit, v, st); e.record(st)
    torch.cuda.synchronize()
    abi.check_errors(0)
    return float(np.median([s.elapsed_time(e) for s, e in ev])), abi.from_device(hd, F.HIT1)


print(f"{'layout:variant':30s} " + " ".join(f"{k + ' ms':>14s} {'Mrays/s':>9s}"
    for k in sets) + "   any-hit: " + " ".join(f"{k + ' ms':>14s}" for k in sets))
for width in [int(x) for x in a.widths.split(",")]:
    bvh = abi.DeviceBvh.load(path, width, 0)
    names = abi.variants(width)
    base = {}
    for v in (range(len(names)) if a.all_variants else [0]):
        if a.only and v != 0 and names[v] not in a.only.split(","):
            continue
        cols, anycols, ok = [], [], True
        for k in sets:
            ms, h = timed(bvh, k, False, v)
            cols += [ms, len(sets[k]) / ms / 1e3]
            base.setdefault(k, h)
            ok &= h.tobytes() == base[k].tobytes()
            ms_any, ha = timed(bvh, k, True, v)
            anycols.append(ms_any)
            ok &= bool(((ha["tri_id"] >= 0) == (base[k]["tri_id"] >= 0)).all())
        print(f"BVH{width}:{names[v]:24s} " + " ".join(f"{c:14.4f}" if i % 2 == 0 else f"{c:9.1f}" for i,
            c in enumerate(cols)) + "            " +
              " ".join(f"{c:14.4f}" for c in anycols) + ("" if ok else "   RESULTS DIFFER"), flush=True)
