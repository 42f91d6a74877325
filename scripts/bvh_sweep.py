#!/usr/bin/env python
"""The renderer's own BVH for this GPU: `rodent` / converter build the hierarchy themselves (host/bvh_build.cpp), unlike
bench_traversal, which is handed one.  Sweeps the builder's leaf threshold (the reference stops at 2 references, converter.cpp:144) and the
traversal-cost constant (the reference: 1 x half area, converter.cpp:120-127, bvh.h:172-176) on the atrium: nodes, references, the
oracle's inner-node / triangle visits per ray (camera rays and the benchmark's random segments), and the frame rate of config 5's frame.
Every frame must have the ray counts of the first one (the same paths whatever the hierarchy).
usage: python scripts/bvh_sweep.py [--spp 32] [--leaf 1,2,3,4,6,8] [--ct 0.5,1,2,3]"""
import argparse, subprocess, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import binding as O
from rodent_amd import build, formats as F, raygen, render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--leaf", default="1,2,3,4,6,8")
ap.add_argument("--ct", default="0.5,1,2,3")
ap.add_argument("--frames", type=int, default=2)
a = ap.parse_args()
w, h = (int(x) for x in a.size.split("x"))
scenes.scene_bvh("atrium")
obj = scenes.DATA / "atrium.obj"
eye, d, up, fov = scenes.CAMERAS["atrium"]
cam = S.camera_settings(eye, d, up, fov, w, h)
prim = raygen.primary_rays(eye, d, up, fov, 512, 512, 0.0, 5000.0)
rnd = None
first_counts = None
print(f"atrium, {w}x{h} x {a.spp} spp, path length 8; oracle visits per ray on 512 x 512 camera rays and 256 Ki random segments")
print(f"{'leaf':>4s} {'Ct':>4s} | {'nodes':>7s} {'refs':>7s} | {'camera: inner':>13s} {'tris':>6s} {'steps':>6s} | {'random: inner':>13s} "
    f"{'tris':>6s} {'steps':>6s} | {'frame ms':>8s} {'Msamples/s':>10s}")
for leaf in [int(x) for x in a.leaf.split(",")]:
    for ct in [float(x) for x in a.ct.split(",")]:
        out = Path("/tmp") / f"atrium_l{leaf}_c{ct}.rscene"
        subprocess.run([str(build.BIN_DIR / "converter"), str(obj), "-o", str(out), "--bvh-leaf", str(leaf), "--bvh-traversal-cost",
            str(ct)], check=True, stdout=subprocess.DEVNULL)
        sc = S.Scene(out)
        nodes, tris = sc.nodes, sc.tris
        if rnd is None:
            b = np.asarray(nodes["bounds"][0]).reshape(2, 6)
            lo = np.minimum(b[0, 0::2], b[1, 0::2]); hi = np.maximum(b[0, 1::2], b[1, 1::2])
            rnd = raygen.random_rays(lo, hi, 1 << 18, 42, 0.0, 1.0)
        _, sp = O.traverse(2, nodes, tris, prim)
        _, sr = O.traverse(2, nodes, tris, rnd)
        r = R.Renderer(sc, w, h, spp=4, max_path_len=8, dev=0, mapping="auto")
        r.render_rows(cam, 0, 0, h)
        r.configure(a.spp, 8)
        ms = []
        for it in range(a.frames + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.render_rows(cam, it, 0, h)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        c = r.counters(); r.close()
        counts = (c["primary_rays"], c["shadow_rays"])
        first_counts = first_counts or counts
        best = float(np.median(ms[1:]))
        print(f"{leaf:4d} {ct:4.1f} | {len(nodes):7d} {len(tris):7d} | {sp['inner_per_ray']:13.2f} {sp['prims_per_ray']:6.2f} "
            f"{sp['inner_per_ray'] + sp['prims_per_ray']:6.2f} | "
              f"{sr['inner_per_ray']:13.2f} {sr['prims_per_ray']:6.2f} {sr['inner_per_ray'] + sr['prims_per_ray']:6.2f} | {best:8.1f} "
                  f"{a.spp * w * h / best / 1e3:10.1f}" + ("" if counts == first_counts else f"  RAY COUNTS DIFFER {counts}"), flush=True)
