#!/usr/bin/env python
"""Deferred leaves (lab variants "defer-*", csrc/lab/defer_kernels.h) against the shipped default ("top") and "refill":
kernel ms on the benchmark's camera rays and random segments (1 Mi each; --big adds 8 Mi segments), closest and any hit, and -- since a ray
that walks on against a stale tmax visits a superset of the reference's nodes -- the exact number of rays whose Hit1 record differs from the
default's (= the oracle's, checked once): ids that differ, t that differs and by how many ulps.
usage: RODENT_HIP_LAB=1 python scripts/defer_experiment.py [--scene atrium] [--steps 20] [--big] [--only PATTERN]"""
import argparse, re, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="atrium")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--big", action="store_true")
ap.add_argument("--only", default="defer")
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()

path = scenes.scene_bvh(a.scene)
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS[a.scene]
lo, hi = raygen.scene_bounds(F.read_bvh(path, F.BVH4_TRI4)[0])
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
if a.big:
    sets["random8Mi"] = raygen.random_rays(lo, hi, 1 << 23, 43, 0.0, 1.0)
names = abi.variants(2)
todo = [i for i, nm in enumerate(names) if nm in ("top", "refill") or re.search(a.only, nm)]


def run(v, rays, any_hit):
    n = len(rays)
    rd = abi.to_device(rays, 0)
    hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, any_hit, v, st)
    torch.cuda.synchronize()
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    first.record(st)
    for _ in range(a.steps):
        abi.traverse_async(bvh, rd, hd, n, any_hit, v, st)
    last.record(st)
    torch.cuda.synchronize()
    abi.check_errors(0)
    return first.elapsed_time(last) / a.steps, abi.from_device(hd, F.HIT1).copy()


def ulps(x, y):
    xi, yi = x.view(np.int32).astype(np.int64), y.view(np.int32).astype(np.int64)
    return np.abs(xi - yi)


base = {}
print(f"scene {a.scene}; ms = mean of {a.steps} back-to-back launches (HIP events); diff = rays whose record differs from the default's: "
    f"ids / t (max ulps)")
print(f"{'variant':22s} " + " ".join(f"{k + (' any' if any_hit else ''):>30s}" for k in sets for any_hit in (False, True)))
for v in todo:
    cells = []
    for k, rays in sets.items():
        for any_hit in (False, True):
            ms, h = run(v, rays, any_hit)
            key = (k, any_hit)
            if key not in base:
                base[key] = h
                if not a.no_oracle and not any_hit and len(rays) <= (1 << 20):
                    from oracle import binding as O
                    nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
                    ref, _ = O.traverse(2, nodes, tris, rays)
                    assert h.tobytes() == ref.tobytes(), "the default differs from the oracle"
            b = base[key]
            if any_hit:
                wrong = int(((h["tri_id"] >= 0) != (b["tri_id"] >= 0)).sum())
                rec = int((h.view(np.uint8).reshape(-1, 16) != b.view(np.uint8).reshape(-1, 16)).any(axis=1).sum())
                cells.append(f"{ms:8.4f} occl {wrong} rec {rec}")
            else:
                ids = int((h["tri_id"] != b["tri_id"]).sum())
                dt = ulps(h["t"], b["t"])
                cells.append(f"{ms:8.4f} ids {ids} t {int((dt > 0).sum())} ({int(dt.max())})")
    print(f"{names[v]:22s} " + " ".join(f"{c:>30s}" for c in cells), flush=True)
# what the instrumented builds counted: drain rounds and lane-rounds (stats[3], stats[4]) -> lane utilisation of the triangle rounds
for v in [i for i, nm in enumerate(names) if nm.startswith("stats-defer")]:
    for k in ("primary", "random"):
        abi.read_stats(0)
        rays = sets[k]
        abi.traverse(bvh, rays, variant=v)
        st = abi.read_stats(0)
        rounds, lanes = int(st[3]), int(st[4])
        print(f"{names[v]:22s} {k:8s} drain rounds per ray {rounds * 64 / len(rays):.2f} (wave rounds {rounds}), lanes busy in a round "
            f"{lanes / max(1, rounds) / 64:.3f}, triangle tests per ray {lanes / len(rays):.3f}")
