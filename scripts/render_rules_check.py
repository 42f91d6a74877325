#!/usr/bin/env python
"""The renderer's per-scene rules (include/rodent_render.h: megakernel up to 128 nodes, joint persistent launch above, lane refill 40 / 40
from 16 384 nodes) on a scene 16 x the size of the one they were fitted on: the gallery (the atrium at 4.2 M triangles, 2.3 M nodes) at 1920
x 1080, path length 8 -- what the library chooses (auto) against the choices it did not make.  Also RODENT_HIP_SHADOW_ORDER 0 / 1 / 2 on the
atrium at config 5's size: any-hit rays need no near-first order; films must agree (occlusion does not depend on the
order). usage: python scripts/render_rules_check.py [--spp 16] [--frames 2] [--scene gallery]"""
import argparse, os, subprocess, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="gallery"); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--frames", type=int,
    default=2)
ap.add_argument("--shadow-order-child", type=int, default=-1,
    help="internal: one RODENT_HIP_SHADOW_ORDER run (the value is read once per process)")
a = ap.parse_args()
import torch
import bench
from rodent_amd import render as R, scene as S, scenes


def frames(sc, cam, w, h, spp, n, **opts):
    r = R.Renderer(sc, w, h, spp=4, max_path_len=8, dev=0, **opts)
    r.render_rows(cam, 0, 0, h)
    r.configure(spp, 8); r.clear()
    ms = []
    for it in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_rows(cam, it, 0, h)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    out = (float(np.median(ms)), r.mapping_name(), r.counters(), r.film().copy())
    r.close()
    return out


if a.shadow_order_child >= 0:
    sc = S.Scene(bench.scene_file("atrium")[1])
    cam = S.camera_settings(*scenes.CAMERAS["atrium"], 3840, 2160)
    ms, name, c, film = frames(sc, cam, 3840, 2160, 32, 3, mapping="auto")
    np.save(f"/tmp/shadow_order_{a.shadow_order_child}.npy", film)
    print(f"RODENT_HIP_SHADOW_ORDER={a.shadow_order_child}: atrium 3840x2160 x 32 spp len 8 ({name}): {ms:.1f} ms = "
        f"{32 * 3840 * 2160 / ms / 1e3:.1f} Msamples/s; rays {c['primary_rays']} + {c['shadow_rays']} shadow", flush=True)
    sys.exit(0)

w, h = 1920, 1080
t0 = time.time()
sc = S.Scene(bench.scene_file(a.scene)[1])
cam = S.camera_settings(*scenes.CAMERAS[a.scene.split("/")[0]], w, h)
print(f"== {a.scene}: {sc.num_tris} triangles, {len(sc.nodes)} nodes ({time.time() - t0:.0f} s to generate / convert / load), {w}x{h} x "
    f"{a.spp} spp, path length 8")
base = None
for label, opts in (("auto (the library's rules)", dict(mapping="auto")), ("megakernel", dict(mapping="megakernel")),
    ("streaming, two streams (not joint)", dict(mapping="streaming", trace_persistent=1)),
                    ("streaming, joint, whole chunks (refill 0)", dict(mapping="streaming", trace_persistent=2, trace_refill=0)),
                    ("streaming, joint, refill 32 / 32", dict(mapping="streaming", trace_persistent=2, trace_refill=32)),
                    ("streaming, joint, refill 48 / 48", dict(mapping="streaming", trace_persistent=2, trace_refill=48)),
                    ("streaming, sorted by material", dict(mapping="streaming", sort=True))):
    ms, name, c, film = frames(sc, cam, w, h, a.spp, a.frames, **opts)
    base = base if base is not None else (c, film)
    same = (c["primary_rays"], c["shadow_rays"]) == (base[0]["primary_rays"],
        base[0]["shadow_rays"]) and bool(np.allclose(film, base[1], rtol=1e-4, atol=1e-5))
    print(f"  {label:44s} {ms:8.1f} ms = {a.spp * w * h / ms / 1e3:7.1f} Msamples/s  ({name})  same counts and film as auto: {same}",
        flush=True)
for mode in (0, 1, 2, 0, 1, 2):
    subprocess.run([sys.executable, __file__, "--shadow-order-child", str(mode)], env=dict(os.environ, RODENT_HIP_SHADOW_ORDER=str(mode)),
        check=True)
f = [np.load(f"/tmp/shadow_order_{m}.npy") for m in (0, 1, 2)]
print("films of the three shadow orders agree (1e-4 relative):",
    bool(np.allclose(f[0], f[1], rtol=1e-4, atol=1e-5) and np.allclose(f[0], f[2], rtol=1e-4, atol=1e-5)))
