#!/usr/bin/env python
"""Where the two renderer mappings cross: the atrium decimated to 1/1 ... 1/512 of its faces (every k-th face kept, all
emitters kept) and the Cornell box, each rendered at 1920 x 1080 x 16 spp, path length 8, streaming against megakernel.
Prints inner BVH nodes, Msamples/s of both and which one the library's per-scene choice (rodent_hip_render_mapping(dev, -1))
takes.  usage: python scripts/mapping_sweep.py [--spp 16]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--keep", default="512,128,32,8,2,1")
a = ap.parse_args()
scenes.scene_bvh("atrium")
src = scenes.DATA / "atrium.obj"
mtl = (scenes.DATA / "atrium.mtl").read_text()
emissive, cur = set(), None
for line in mtl.splitlines():
    t = line.split()
    if t[:1] == ["newmtl"]:
        cur = t[1]
    if t[:1] == ["Ke"] and any(float(x) > 0 for x in t[1:4]):
        emissive.add(cur)


def decimate(keep_every, dst):
    k, mat = 0, None
    with open(src) as f, open(dst, "w") as out:
        for line in f:
            if line.startswith("usemtl"):
                mat = line.split()[1]
            if line.startswith("f "):
                if mat in emissive or k % keep_every == 0:
                    out.write(line)
                k += 1
            else:
                out.write(line)
    return dst


W, H, LEN = 1920, 1080, 8
cases = [("cornell", scenes.GOLDEN / "cornell_box.obj", scenes.CAMERAS["cornell"])]
for k in [int(x) for x in a.keep.split(",")]:
    dst = Path("/tmp") / f"atrium-keep{k}.obj"
    if k > 1:
        (Path("/tmp") / "atrium.mtl").write_text(mtl)
        decimate(k, dst)
    cases.append((f"atrium 1/{k}", dst if k > 1 else src, scenes.CAMERAS["atrium"]))
print(f"{'scene':14s} {'triangles':>9s} {'BVH nodes':>9s} {'streaming':>10s} {'megakernel':>10s}   Msamples/s at {W}x{H}x{a.spp} spp, path "
    f"length {LEN};  library's choice")
for name, obj, (eye, d, up, fov) in cases:
    sc = S.convert(obj, Path("/tmp") / "sweep.rscene")
    cam = S.camera_settings(eye, d, up, fov, W, H)
    rates = {}
    for mapping in ("streaming", "megakernel", "auto"):
        r = R.Renderer(sc, W, H, a.spp, LEN, mapping=mapping)
        if mapping == "auto":
            choice = r.mapping_name(); r.close(); break
        r.render(cam, 0); secs = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_rows(cam, it + 1, 0, H); secs.append(time.perf_counter() - t0)
        rates[mapping] = a.spp * W * H / float(np.median(secs)) / 1e6
        r.close()
    best = max(rates, key=rates.get)
    print(f"{name:14s} {sc.num_tris:9d} {len(sc.nodes):9d} {rates['streaming']:10.1f} {rates['megakernel']:10.1f}   faster: {best:10s} "
        f"chosen: {choice}{'' if best == choice else '   <-- not the faster one'}", flush=True)
