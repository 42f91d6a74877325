#!/usr/bin/env python
"""What spatially ordered bounce rays would be worth to the renderer's traversal passes.  The streaming loop is driven by hand
through the stage entry points on the atrium (4 Mi paths); after every shader run the stream of continuing rays and the
stream of shadow rays are copied out as Ray1 arrays and traced with the benchmark kernel (closest hit / any hit)
  as they are (stream order = by material of the vertex they leave, pixel order inside a material),
  sorted by the Morton cell of their origin with 2 / 4 / 8 / 16 cells per axis (8 ... 4096 cells; stable, on the host).
Prints Mrays/s per bounce and the sum over the frame.  usage: python scripts/bounce_coherence_experiment.py"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, render as R, scene as S, scenes

scenes.scene_bvh("atrium")
sc = S.convert(scenes.DATA / "atrium.obj", Path("/tmp") / "bce.rscene")
W, H, SPP, MAXLEN = 1024, 512, 8, 8
eye, d, up, fov = scenes.CAMERAS["atrium"]
cam = S.camera_settings(eye, d, up, fov, W, H)
r = R.Renderer(sc, W, H, SPP, MAXLEN, mapping="streaming", sort=False)
l = R.stage_lib()
cap = W * H * SPP
p, q, s = R.PrimaryStream(), R.PrimaryStream(), R.SecondaryStream()
l.rodent_gpu_get_first_primary_stream(0, C.byref(p), cap)
l.rodent_gpu_get_second_primary_stream(0, C.byref(q), cap)
l.rodent_gpu_get_secondary_stream(0, C.byref(s), cap)
st = R.make_settings(cam)
bvh = abi.DeviceBvh(2, sc.nodes, sc.tris, 0)
lo = np.minimum(sc.nodes["bounds"][0][[0, 2, 4]], sc.nodes["bounds"][0][[6, 8, 10]]) if False else sc.vertices[:, :3].min(0)
hi = sc.vertices[:, :3].max(0)
stream = torch.cuda.current_stream()
G = len(sc.materials)
ends = (C.c_int32 * (G + 1))()


def rays_of(rs, n, keep):
    a = np.zeros(n, F.RAY1)
    for k, name in enumerate(("org_x", "org_y", "org_z")):
        a["org"][:, k] = R.read_stream_array(getattr(rs, name), n, "<f4")
    for k, name in enumerate(("dir_x", "dir_y", "dir_z")):
        a["dir"][:, k] = R.read_stream_array(getattr(rs, name), n, "<f4")
    a["tmin"] = R.read_stream_array(rs.tmin, n, "<f4"); a["tmax"] = R.read_stream_array(rs.tmax, n, "<f4")
    return a[keep]


def timed(rays, any_hit):
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        abi.traverse_async(bvh, rd, hd, n, any_hit, 0, stream)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in ev:
        a.record(stream); abi.traverse_async(bvh, rd, hd, n, any_hit, 0, stream); b.record(stream)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def morton_order(org, cells):
    c = np.clip(((org - lo) / np.maximum(hi - lo, 1e-9) * cells).astype(np.int64), 0, cells - 1)
    key = np.zeros(len(org), np.int64)
    for b in range(int(np.log2(cells))):
        for ax in range(3):
            key |= ((c[:, ax] >> b) & 1) << (3 * b + ax)
    return np.argsort(key, kind="stable")


l.hip_generate_rays(0, C.byref(p), cap, 0, cap, C.byref(st), 0, W, H, 0, SPP, None)
totals = {}
print(f"{'bounce':>6s} {'kind':>8s} {'rays':>9s} " + " ".join(f"{c:>12s}" for c in ("stream order", "8 cells", "64 cells", "512 cells",
    "4096 cells")) + "   ms per pass")
bounce = 0
while p.size > 0 and bounce < MAXLEN + 1:
    l.hip_traverse_primary(0, C.byref(p), None)
    # the stage-level shader wants the misses dropped (mapping_gpu.impala:347-357)
    l.hip_sort_primary(0, C.byref(p), C.byref(q), ends, None)
    p, q = q, p
    n = ends[G - 1]
    l.hip_shade(0, C.byref(p), C.byref(s), n, None)
    ids = R.read_stream_array(p.rays.id, n, "<i4"); sids = R.read_stream_array(s.rays.id, n, "<i4")
    for kind, rs, keep, any_hit in (("shadow", s.rays, sids >= 0, True), ("bounce", p.rays, ids >= 0, False)):
        rays = rays_of(rs, n, keep)
        if len(rays) < 1 << 16:
            continue
        row = [timed(rays, any_hit)] + [timed(rays[morton_order(rays["org"], c)], any_hit) for c in (2, 4, 8, 16)]
        for k, v in enumerate(row):
            totals.setdefault(kind, [0.0] * 5)[k] += v
        print(f"{bounce:6d} {kind:>8s} {len(rays):9d} " + " ".join(f"{v:12.4f}" for v in row), flush=True)
    l.hip_traverse_secondary(0, C.byref(s), None)
    l.hip_compact_primary(0, C.byref(p), C.byref(q), None)
    p, q = q, p
    bounce += 1
for kind, t in totals.items():
    print(f"{'sum':>6s} {kind:>8s} {'':9s} " + " ".join(f"{v:12.4f}"
        for v in t) + "   ratio to stream order: " + " ".join(f"{t[0] / v:.3f}" for v in t))
r.close()
