#!/usr/bin/env python
"""Wave-cooperative BVH8 for any-hit rays (lab variant "coop" of the BVH8 table: one ray per octet of lanes, csrc/lab/coop8_kernel.h;
VERDICT r5 item 6) against the one-ray-per-lane BVH8 default and the BVH2 default: 1 Mi camera rays, ao rays (ray_gen's shadow mode) and
random segments, any hit; ms per launch (30 launches, best of 3). Parity: the cooperative kernel's Hit1 records against the one-ray-per-lane
kernel's (the same visit order: bit for bit) and its occlusion answers against the oracle's (B1g). usage: RODENT_HIP_LAB=1 python
scripts/coop8_experiment.py [scene]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
path = scenes.scene_bvh(scene)
eye, d, up, fov = scenes.CAMERAS[scene.split("/")[0]]
st = torch.cuda.current_stream()
n2, t2 = F.read_bvh(path, F.BVH2_TRI1)
n8, t8 = F.read_bvh(path, F.BVH8_TRI4)
lo, hi = raygen.scene_bounds2(n2)
prim = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX)
bvh2, bvh8 = abi.DeviceBvh(2, n2, t2, 0), abi.DeviceBvh(8, n8, t8, 0)
hits = abi.traverse(bvh2, prim)
sets = {"camera": prim, "ao": raygen.shadow_rays(scenes.LIGHTS[scene.split("/")[0]], prim, hits["t"], 0.0, 0.999),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, scenes.RANDOM_TMAX)}
coop = abi.variants(8).index("coop")


def timed(bvh, rd, hd, n, variant):
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, True, variant, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            abi.traverse_async(bvh, rd, hd, n, True, variant, st)
        e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    return best, abi.from_device(hd, F.HIT1).copy()


print(f"== {scene}: any-hit rays, ms per launch of 1 Mi rays                BVH2 default   BVH8 default   BVH8 cooperative   records = BVH8 "
    f"default's   occlusion = oracle's (64 Ki sample)")
for name, rays in sets.items():
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    m2, _ = timed(bvh2, rd, hd, n, 0)
    m8, h8 = timed(bvh8, rd, hd, n, 0)
    hd.zero_()
    mc, hc = timed(bvh8, rd, hd, n, coop)
    from oracle import binding as O
    ref, _ = O.traverse(8, n8, t8, rays[:1 << 16], algo="gpu", any_hit=True)
    occl = bool(((hc[:1 << 16]["tri_id"] >= 0) == (ref["tri_id"] >= 0)).all())
    print(f"   {name:8s}                                                     {m2:8.4f}       {m8:8.4f}       {mc:8.4f}           "
        f"{hc.tobytes() == h8.tobytes()}                      {occl} (records {hc[:1 << 16].tobytes() == ref.tobytes()})", flush=True)
abi.check_errors(0)
print("deep rays handed to the follow-up kernel in the last launch:", int(abi.read_stats(0)[7]))
