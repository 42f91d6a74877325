#!/usr/bin/env python
"""Chunk SHAPE experiment (lab build): the primary set of the benchmark camera is a 1024 x 1024 (or 4096 x 4096) image in
scanline order, so a 64-ray chunk is a 64 x 1 pixel strip.  The same rays through "top-userperm" with every chunk a
th x tw pixel tile instead (tiles in row-major order, or in Morton order so that a stripe's 32-chunk group is a compact
block of the image), and -- for the upper bound -- tiles with each stripe's chunks longest-first by the oracle's cost.
Hits go to hits[ray index]: results are identical by construction (checked).
Prints the model (wave iterations per chunk from the oracle's per-ray step counts) next to the measured time.
usage: RODENT_HIP_LAB=1 python scripts/tile_experiment.py [--big] [--kernel top-userperm]"""
import argparse
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes
from oracle import binding as O

ap = argparse.ArgumentParser()
ap.add_argument("--big", action="store_true", help="also the 4096 x 4096 image (16 Mi rays per launch)")
ap.add_argument("--kernel", default="top-userperm")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
eye, d, up, fov = scenes.CAMERAS["atrium"]
names = abi.variants(2)
STRIPES, GROUP = 64, 32


def timed(v, rd, hd, n, steps):
    st = torch.cuda.current_stream()
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


def morton_order(rows, cols):
    """Tile indices (row-major ids) of a rows x cols tile grid in Morton order."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    key = np.zeros(r.shape, np.int64)
    for b in range(12):
        key |= ((c >> b) & 1) << (2 * b)
        key |= ((r >> b) & 1) << (2 * b + 1)
    return np.argsort(key.ravel(), kind="stable")


def tile_perm(w, h, th, tw, order="row"):
    """perm[chunk * 64 + j] = ray index: chunk = tile (th x tw pixels), tiles in row-major or Morton order."""
    rows, cols = h // th, w // tw
    tiles = np.arange(rows * cols) if order == "row" else morton_order(rows, cols)
    tr, tc = tiles // cols, tiles % cols
    jy, jx = np.arange(th * tw) // tw, np.arange(th * tw) % tw
    return (((tr[:, None] * th + jy[None, :]) * w) + tc[:, None] * tw + jx[None, :]).astype(np.int32).ravel()


def lpt(perm, cost_ray, n):
    """Reorders the chunks of `perm` so that every stripe draws its chunks longest first (oracle cost)."""
    chunks = n // 64
    cost = cost_ray[perm].reshape(chunks, 64).max(1)
    per_stripe = chunks // STRIPES
    pos = np.array([[((t // GROUP) * STRIPES + s) * GROUP + t % GROUP for t in range(per_stripe)] for s in range(STRIPES)])
    out = perm.reshape(chunks, 64).copy()
    src = perm.reshape(chunks, 64)
    for s in range(STRIPES):
        c = pos[s]
        out[c] = src[c[np.argsort(-cost[c], kind="stable")]]
    return out.ravel()


sizes = [(1024, 1024)] + ([(4096, 4096)] if a.big else [])
for w, h in sizes:
    rays = raygen.primary_rays(eye, d, up, fov, w, h, 0.0, 5000.0)
    n = len(rays)
    cost_ray = O.ray_steps(nodes, tris, rays).sum(1)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    steps = a.steps if n <= (1 << 21) else 8
    base_ms = timed(names.index("top"), rd, hd, n, steps)
    ref = abi.from_device(hd, F.HIT1).tobytes()
    it0 = cost_ray.reshape(-1, 64).max(1)
    print(f"{w}x{h}: default kernel, scanline chunks: {base_ms:.4f} ms = {n / base_ms / 1e3:.0f} Mrays/s; model: {it0.mean():.2f} "
        f"iterations per chunk, lane utilisation {cost_ray.sum() / (it0.sum() * 64):.3f}")
    cases = [("identity through the permutation", np.arange(n, dtype=np.int32))]
    for th, tw in ((2, 32), (4, 16), (8, 8), (16, 4)):
        cases.append((f"{th}x{tw} tiles, row-major", tile_perm(w, h, th, tw)))
    cases.append(("8x8 tiles, Morton order", tile_perm(w, h, 8, 8, "morton")))
    cases.append(("4x16 tiles, Morton order", tile_perm(w, h, 4, 16, "morton")))
    if n <= (1 << 20):
        cases.append(("scanline chunks, longest first per stripe (oracle cost)", lpt(np.arange(n, dtype=np.int32), cost_ray, n)))
        cases.append(("8x8 tiles row-major, longest first per stripe (oracle cost)", lpt(tile_perm(w, h, 8, 8), cost_ray, n)))
    v = names.index(a.kernel)
    for label, perm in cases:
        assert np.array_equal(np.sort(perm), np.arange(n))
        pd = torch.from_numpy(perm).cuda()
        abi.lib().rodent_hip_debug_set_perm(0, pd.data_ptr())
        ms = timed(v, rd, hd, n, steps)
        same = abi.from_device(hd, F.HIT1).tobytes() == ref
        it = cost_ray[perm].reshape(-1, 64).max(1)
        print(f"   {label:62s} {ms:.4f} ms ({base_ms / ms:.3f} x) {n / ms / 1e3:7.0f} Mrays/s  identical {same}   model: {it.mean():.2f} "
            f"it/chunk, util {cost_ray.sum() / (it.sum() * 64):.3f}", flush=True)
    abi.lib().rodent_hip_debug_set_perm(0, None)
    del rd, hd
