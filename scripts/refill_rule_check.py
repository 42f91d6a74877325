#!/usr/bin/env python
"""The renderer's lane-refill thresholds (include/rodent_render.h: 40 / 40 from 16 384 nodes, fitted on the atrium) on all four scene
classes: atrium, gallery, crown, plant through the streaming mapping with the joint launch, thresholds (bounce : shadow) swept; ray counts
must agree exactly with the library's choice, films up to the order of the atomic adds.  Each cell is timed twice (in two passes over the
table) and the better median counts. usage: python scripts/refill_rule_check.py [--size 3840x2160] [--spp 16] [--frames 3] [--scenes
atrium,gallery,crown,plant] [--idle 24,32,40,48,32:40,40:32]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from rodent_amd import render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160"); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--frames", type=int,
    default=3)
ap.add_argument("--scenes", default="atrium,gallery,crown,plant"); ap.add_argument("--idle", default="24,32,40,48,32:40,40:32")
a = ap.parse_args()
w, h = (int(x) for x in a.size.split("x"))
idle = [tuple(int(v) for v in x.split(":")) if ":" in x else (int(x), int(x)) for x in a.idle.split(",")]


def frames(sc, cam, **opts):
    r = R.Renderer(sc, w, h, spp=4, max_path_len=8, dev=0, **opts)
    r.render_rows(cam, 0, 0, h)
    r.configure(a.spp, 8); r.clear()
    ms = []
    for it in range(a.frames):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_rows(cam, it, 0, h)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    out = (float(np.median(ms)), r.mapping_name(), r.trace_refill(), r.counters(), r.film().copy())
    r.close()
    return out


for name in a.scenes.split(","):
    t0 = time.time()
    sc = S.Scene(bench.scene_file(name)[1])
    cam = S.camera_settings(*scenes.CAMERAS[name.split("/")[0]], w, h)
    print(f"== {name}: {sc.num_tris} triangles, {len(sc.nodes)} nodes ({time.time() - t0:.0f} s to generate / convert / load), {w}x{h} x "
        f"{a.spp} spp, path length 8", flush=True)
    cells = [("auto", dict(mapping="auto"))] + [(f"{b} : {s}", dict(mapping="streaming", trace_persistent=2, trace_refill=(b, s))) for b,
        s in idle]
    best, base, chosen = {}, None, None
    for _ in range(2):
        for label, opts in cells:
            ms, mname, refill, c, film = frames(sc, cam, **opts)
            if label == "auto":
                chosen = (mname, refill)
                base = base if base is not None else (c, film)
            same = (c["primary_rays"], c["shadow_rays"]) == (base[0]["primary_rays"],
                base[0]["shadow_rays"]) and bool(np.allclose(film, base[1], rtol=1e-4, atol=1e-5))
            best[label] = (min(ms, best[label][0]) if label in best else ms, same and best.get(label, (0, True))[1])
    ref = best["auto"][0]
    print(f"   the library chooses {chosen[0]}, refill {chosen[1][0]} : {chosen[1][1]}")
    for label, _ in cells:
        ms, same = best[label]
        print(f"   {label:>8s}  {ms:8.1f} ms = {a.spp * w * h / ms / 1e3:7.1f} Msamples/s  ({ref / ms:5.3f} x auto)  same counts and film: "
            f"{same}", flush=True)
