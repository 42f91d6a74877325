#!/usr/bin/env python
"""What a better chunk ORDER could buy the persistent default kernel at 1 Mi rays (lab build): the same rays through
"top-userperm" with the chunks of every stripe re-ordered by their cost (wave iterations = max over the chunk's 64 rays of the
oracle's step counts): longest first over the whole stripe; first generation as it is, drawn chunks longest first; and drawn
chunks ordered by the cost of their vertical neighbour in the first generation.  Results are identical by construction.
usage: RODENT_HIP_LAB=1 python scripts/lpt_experiment.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes
from oracle import binding as O

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
names = abi.variants(2)
STRIPES, GROUP = 64, 32


def position(stripe, t):                     # chunk that ticket t of a stripe traces in the default order
    return ((t // GROUP) * STRIPES + stripe) * GROUP + t % GROUP


def timed(v, rd, hd, n, steps=20):
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


for k, rays in sets.items():
    n = len(rays); chunks = n // 64; per_stripe = chunks // STRIPES
    cost = O.ray_steps(nodes, tris, rays).sum(1).reshape(-1, 64).max(1)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    base_ms = timed(names.index("top"), rd, hd, n)
    ref = abi.from_device(hd, F.HIT1).tobytes()
    print(f"{k}: default order {base_ms:.4f} ms")
    orders = {}
    pos = np.array([[position(s, t) for t in range(per_stripe)] for s in range(STRIPES)])       # [stripe][ticket] -> chunk
    ident = pos.copy(); orders["identity through the permutation"] = ident
    full = pos.copy(); second = pos.copy(); neigh = pos.copy()
    half = per_stripe // 2
    for s in range(STRIPES):
        c = pos[s]
        full[s] = c[np.argsort(-cost[c], kind="stable")]
        second[s, half:] = c[half:][np.argsort(-cost[c[half:]], kind="stable")]
    groups = per_stripe // GROUP
    # even 32-chunk groups of the stripe first, then the odd ones
    inter = np.concatenate([np.arange(0, groups, 2), np.arange(1, groups, 2)])
    orders["even groups first, then odd groups (stateless)"] = np.stack([pos[s].reshape(groups, GROUP)[inter].ravel()
        for s in range(STRIPES)])
    rev = np.arange(groups)[::-1]
    orders["groups in reverse order (stateless)"] = np.stack([pos[s].reshape(groups, GROUP)[rev].ravel() for s in range(STRIPES)])
    orders["longest first (whole stripe)"] = full
    orders["first generation unchanged, drawn chunks longest first"] = second
    for name, order in orders.items():
        perm = np.empty(n, np.int32)
        for s in range(STRIPES):
            for t in range(per_stripe):
                p, c = pos[s, t], order[s, t]
                perm[p * 64:(p + 1) * 64] = np.arange(c * 64, (c + 1) * 64)
        assert np.array_equal(np.sort(perm), np.arange(n))
        pd = torch.from_numpy(perm).cuda()
        abi.lib().rodent_hip_debug_set_perm(0, pd.data_ptr())
        ms = timed(names.index("top-userperm"), rd, hd, n)
        same = abi.from_device(hd, F.HIT1).tobytes() == ref
        print(f"   {name:60s} {ms:.4f} ms ({base_ms / ms:.3f} x)  identical {same}")
    abi.lib().rodent_hip_debug_set_perm(0, None)
