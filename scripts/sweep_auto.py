#!/usr/bin/env python
"""The default mapping (k_bvh2_top_auto: whole chunks for rays that share an origin or a direction, lane refill otherwise) against the two
kernels it chooses between -- "top-chunks" (lab build: round 3's default, whole chunks only) and "refill" -- on the benchmark's ray sets at
1 Mi rays per launch and at 16 Mi primary / 8 Mi random rays, closest and any hit.  Hits must be identical.
usage: RODENT_HIP_LAB=1 python scripts/sweep_auto.py [--steps 30] [--small]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--small", action="store_true", help="1 Mi-ray sets only")
ap.add_argument("--variants", default="top,top-nohint,top-chunks,refill",
    help="top-nohint = the default with rodent_hip_ray_kind_hint(0): the in-kernel choice alone")
a = ap.parse_args()

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary 1Mi": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
        "random 1Mi": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
if not a.small:
    sets["primary 16Mi"] = raygen.primary_rays(eye, d, up, fov, 4096, 4096, 0.0, 5000.0)
    sets["random 8Mi"] = raygen.random_rays(lo, hi, 1 << 23, 42, 0.0, 1.0)
names = abi.variants(2)
todo = [(v, names.index(v.replace("-nohint", ""))) for v in a.variants.split(",") if v.replace("-nohint", "") in names]
for any_hit in (False, True):
    base = {}
    print(f"{'any hit' if any_hit else 'closest hit':28s} " + " ".join(f"{k + ' ms':>16s}" for k in sets) + "  identical")
    for label, v in todo:
        abi.ray_kind_hint(not label.endswith("-nohint"))
        row, same = [], []
        for k, rays in sets.items():
            n = len(rays)
            rd = abi.to_device(rays, 0)
            hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
            st = torch.cuda.current_stream()
            for _ in range(3):
                abi.traverse_async(bvh, rd, hd, n, any_hit, v, st)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
            for s, e in ev:
                s.record(st); abi.traverse_async(bvh, rd, hd, n, any_hit, v, st); e.record(st)
            torch.cuda.synchronize()
            row.append(float(np.median([s.elapsed_time(e) for s, e in ev])))
            h = abi.from_device(hd, F.HIT1)
            base.setdefault(k, h)
            if label.startswith("steal") and not any_hit:      # order-changing: how many rays differ from the first row (t bits / ids)
                same.append((int((h["t"].view("<u4") != base[k]["t"].view("<u4")).sum()), int((h["tri_id"] != base[k]["tri_id"]).sum())))
            else:
                same.append(h.tobytes() == base[k].tobytes() if not any_hit
                    else bool(((h["tri_id"] >= 0) == (base[k]["tri_id"] >= 0)).all()))
            del rd, hd
        print(f"{v}:{label:26s} " + " ".join(f"{x:16.4f}" for x in row) + f"  {same}", flush=True)
