#!/usr/bin/env python
"""Frame rate of one renderer configuration with the library's defaults (A/B runs: environment switches, BVH builder parameters).
usage: python scripts/frame_rate.py [--scene atrium] [--size 3840x2160] [--spp 64] [--len 8] [--frames 3] [--rscene file]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from rodent_amd import render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="atrium")
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--len", type=int, default=8)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--rscene", default=None)
ap.add_argument("--mapping", default="auto")
a = ap.parse_args()
w, h = (int(x) for x in a.size.split("x"))
rscene = a.rscene or bench.scene_file(a.scene)[1]
sc = S.Scene(rscene)
eye, d, up, fov = scenes.CAMERAS[a.scene.split("/")[0]]
cam = S.camera_settings(eye, d, up, fov, w, h)
r = R.Renderer(sc, w, h, spp=4, max_path_len=a.len, dev=0, mapping=a.mapping)
r.render_rows(cam, 0, 0, h)
r.configure(a.spp, a.len)
ms = []
for it in range(a.frames + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render_rows(cam, it, 0, h)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
c = r.counters()
best = float(np.median(ms[1:]))
print(f"{Path(str(rscene)).name} {len(sc.nodes)} nodes {w}x{h} x {a.spp} spp len {a.len} ({r.mapping_name()}): {best:.1f} ms = "
    f"{a.spp * w * h / best / 1e3:.1f} Msamples/s; rays {c['primary_rays']} + {c['shadow_rays']} shadow", flush=True)
r.close()
