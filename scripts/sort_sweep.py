#!/usr/bin/env python
"""Does the sort by material pay (streaming mapping)?  Sorted against unsorted shading at 1920 x 1080 x 16 spp on
  cornell     one BSDF kind (diffuse), 4 materials
  atrium      diffuse + diffuse / Phong mixes, 9 materials
  materials   the tests' room with every BSDF kind (diffuse, Phong, mix, mirror, glass, black, emitter) on neighbouring walls
  textured    the tests' textured room (map_Kd PNG / JPEG, map_Ks TGA)
usage: python scripts/sort_sweep.py"""
import sys, time, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import torch
from rodent_amd import render as R, scene as S, scenes
import conftest

tmp = Path(tempfile.mkdtemp())
scenes.scene_bvh("atrium")
cases = [("cornell", scenes.GOLDEN / "cornell_box.obj", ((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60.0), 4),
         ("atrium", scenes.DATA / "atrium.obj", scenes.CAMERAS["atrium"], 8),
         ("materials", conftest.write_materials_scene(tmp / "m"), ((0, 1, 0.9), (0, 0, -1), (0, 1, 0), 75.0), 12),
         ("textured", conftest.write_textured_scene(tmp / "t"), ((0.3, 1.0, 3.2), (-0.1, -0.25, -1), (0, 1, 0), 50.0), 8)]
W, H, SPP = 1920, 1080, 16
print(f"{'scene':10s} {'materials':>9s} {'sorted':>9s} {'unsorted':>9s}   Msamples/s, streaming mapping, {W}x{H}x{SPP} spp")
for name, obj, (eye, d, up, fov), max_len in cases:
    sc = S.convert(obj, tmp / f"{name}.rscene")
    cam = S.camera_settings(eye, d, up, fov, W, H)
    rates = {}
    for sort in (True, False):
        r = R.Renderer(sc, W, H, SPP, max_len, mapping="streaming", sort=sort)
        r.render(cam, 0); secs = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_rows(cam, it + 1, 0, H); secs.append(time.perf_counter() - t0)
        rates[sort] = SPP * W * H / float(np.median(secs)) / 1e6
        r.close()
    print(f"{name:10s} {len(sc.materials):9d} {rates[True]:9.1f} {rates[False]:9.1f}   unsorted / sorted = {rates[False] / rates[True]:.3f}", flush=True)
