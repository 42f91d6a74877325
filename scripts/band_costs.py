#!/usr/bin/env python
"""Predicts the multi-GPU frame time of BASELINE config 5 on ONE GPU: every row band a rank of a 2 / 4 / 8 GPU job would
render (parallel.row_band = host/partition.h split_range) is rendered alone with rodent_hip_render_rows and timed; a frame takes as long as
its slowest band, so efficiency = mean / max and predicted Msamples/s = samples / max.  With --tiles the same for interleaved row tiles
(rodent_hip_render_tiles: rank r renders tiles r, r + N, ... of --tile-rows rows each; reference tile arithmetic
render/mapping_gpu.impala:374-420). usage: python scripts/band_costs.py [--spp 256] [--tiles] [--tile-rows 16] [--gpus 2,4,8]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from rodent_amd import parallel, render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--gpus", default="2,4,8")
ap.add_argument("--tiles", action="store_true")
ap.add_argument("--tile-rows", type=int, default=16)
ap.add_argument("--size", default="3840x2160")
a = ap.parse_args()
w, h = (int(x) for x in a.size.split("x"))
obj, rscene = bench.scene_file("atrium")
sc = S.Scene(rscene)
eye, d, up, fov = scenes.CAMERAS["atrium"]
cam = S.camera_settings(eye, d, up, fov, w, h)
r = R.Renderer(sc, w, h, spp=4, max_path_len=8, dev=0, mapping="auto")
r.render_rows(cam, 0, 0, h)
r.configure(a.spp, 8)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


whole = timed(lambda: r.render_rows(cam, 0, 0, h))
samples = a.spp * w * h
print(f"atrium {w}x{h} x {a.spp} spp, path length 8, mapping {r.mapping_name()}: whole frame on one GPU {whole:.1f} ms = "
    f"{samples / whole / 1e3:.1f} Msamples/s")
print(f"{'GPUs':>4s} {'partition':>22s} | per-rank ms" + " " * 58
    + "| mean / max = efficiency | predicted Msamples/s (film gather not included)")
for n in [int(x) for x in a.gpus.split(",")]:
    for kind in (["bands"] + (["tiles"] if a.tiles else [])):
        ms = []
        for rank in range(n):
            if kind == "bands":
                y0, y1 = parallel.row_band(h, rank, n)
                ms.append(timed(lambda: r.render_rows(cam, 0, y0, y1)))
            else:
                ms.append(timed(lambda: r.render_tiles(cam, 0, a.tile_rows, rank, n)))
        label = f"{h // n}-row bands" if kind == "bands" else f"{a.tile_rows}-row tiles, stride {n}"
        print(f"{n:4d} {label:>22s} | " + " ".join(f"{x:7.1f}"
            for x in ms).ljust(69) + f"| {np.mean(ms):7.1f} / {max(ms):7.1f} = {np.mean(ms) / max(ms):.3f}   | "
            f"{samples / max(ms) / 1e3:8.1f}  ({samples / max(ms) / 1e3 / (samples / whole / 1e3):.2f} x one GPU)", flush=True)
r.close()
