#!/usr/bin/env python
"""Where the default mapping should switch from the one-chunk kernel to the persistent LDS-image kernel, and whether the tuned
constants (switch at 576 Ki rays, 255-record image, 15-entry stack window) hold beyond the scene they were tuned on: both kernels
at 64 Ki ... 2 Mi rays (rodent_hip_top_min_rays(0) forces the persistent kernel) on
  atrium        the benchmark scene (in-tree builder), primary camera + random segments
  refbuilt      tests/golden/atrium-decimated-refbuilt.bvh.gz: every 128th face, hierarchy built by the REFERENCE's bvh.h
  cornell       36 triangles (the whole tree fits the image)
usage: python scripts/threshold_sweep.py [--scenes atrium,refbuilt,cornell]"""
import argparse, gzip, sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="atrium,refbuilt,cornell")
a = ap.parse_args()
names = abi.variants(2)
st = torch.cuda.current_stream()


def timed(bvh, v, rd, hd, n, steps=30):
    for _ in range(4):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


def scene_inputs(name):
    """(bvh path, camera) of a sweep scene."""
    if name == "refbuilt":
        tmp = Path(tempfile.gettempdir()) / "atrium-decimated-refbuilt.bvh"
        tmp.write_bytes(gzip.decompress((scenes.GOLDEN / "atrium-decimated-refbuilt.bvh.gz").read_bytes()))
        return tmp, scenes.CAMERAS["atrium"]
    return scenes.scene_bvh(name), scenes.CAMERAS[name.split("/")[0]]


abi.lib().rodent_hip_top_min_rays(0)
for scene in a.scenes.split(","):
    path, (eye, d, up, fov) = scene_inputs(scene)
    bvh = abi.DeviceBvh.load(path, 2, 0)
    if scene.split("/")[0] in scenes.GENERATED:              # gallery / crown / plant: a .bvh with a BVH2 block only
        lo, hi = raygen.scene_bounds2(F.read_bvh(path, F.BVH2_TRI1)[0])
    else:
        lo, hi = raygen.scene_bounds(F.read_bvh(path, F.BVH4_TRI4)[0])
    print(f"== {scene}: {bvh.num_nodes} nodes, {bvh.num_tris} triangles")
    print(f"{'rays':>10s} {'primary: fast ms':>17s} {'top ms':>9s} {'top/fast':>9s}   {'random: fast ms':>16s} {'top ms':>9s} "
        f"{'top/fast':>9s}")
    for w, h in ((256, 256), (512, 256), (512, 512), (768, 512), (1024, 576), (1024, 768), (1024, 1024), (2048, 1024)):
        rays = raygen.primary_rays(eye, d, up, fov, w, h, 0.0, 5000.0)
        n = len(rays)
        rnd = raygen.random_rays(lo, hi, n, 42, 0.0, 1.0)
        cols = []
        for r in (rays, rnd):
            rd = abi.to_device(r, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
            f, t = timed(bvh, names.index("fast"), rd, hd, n), timed(bvh, names.index("top"), rd, hd, n)
            cols += [f, t, t / f]
        print(f"{n:10d} {cols[0]:17.4f} {cols[1]:9.4f} {cols[2]:9.3f}   {cols[3]:16.4f} {cols[4]:9.4f} {cols[5]:9.3f}", flush=True)
abi.lib().rodent_hip_top_min_rays(-1)
