#!/usr/bin/env python
"""Frame rate of config 5's scene against the streams' capacity (rays per stream; the library's default is 32 Mi).
usage: python scripts/capacity_check.py [--spp 64] [--caps 16,32,64]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from rodent_amd import render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64); ap.add_argument("--caps", default="16,32,64"); ap.add_argument("--scene", default="atrium")
a = ap.parse_args()
w, h = 3840, 2160
sc = S.Scene(bench.scene_file(a.scene)[1])
cam = S.camera_settings(*scenes.CAMERAS[a.scene], w, h)
for rep in range(2):
    for cap in (int(c) << 20 for c in a.caps.split(",")):
        r = R.Renderer(sc, w, h, spp=4, max_path_len=8, dev=0, mapping="auto", capacity=cap)
        r.render_rows(cam, 0, 0, h)
        r.configure(a.spp, 8)
        ms = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.render_rows(cam, it, 0, h)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        c = r.counters()
        best = float(np.median(ms[1:]))
        print(f"capacity {cap >> 20:3d} Mi rays: {best:8.1f} ms = {a.spp * w * h / best / 1e3:7.1f} Msamples/s ({r.mapping_name()}); "
            f"iterations {c['iterations']}", flush=True)
        r.close()
