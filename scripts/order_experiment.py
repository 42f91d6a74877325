#!/usr/bin/env python
"""Chunk ORDER experiment without any cost knowledge (lab build, "top-userperm"): a stripe's first generation of chunks (its
static tickets) is the top half of the image in the default mapping, its second generation the bottom half.  The same rays with
every stripe drawing its 32-chunk groups in another fixed order: reversed, bit-reversed (each generation then samples the whole
image), odd groups first.  Hits identical by construction (checked).
usage: RODENT_HIP_LAB=1 python scripts/order_experiment.py [--steps 30]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--cameras", action="store_true", help="also four other views of the scene and a 4 Mi-ray image")
a = ap.parse_args()
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
names = abi.variants(2)
STRIPES, GROUP = 64, 32


def timed(v, rd, hd, n):
    st = torch.cuda.current_stream()
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


def bitrev(k, bits):
    return int(format(k, f"0{bits}b")[::-1], 2) if bits else 0


def group_perm(n, order):
    """perm over rays: stripe s draws its group slots k = 0, 1, ... in the order order(k, slots)."""
    chunks = n // 64
    slots = chunks // (STRIPES * GROUP)                 # group slots per stripe (8 at 1 Mi rays)
    src = np.arange(n, dtype=np.int32).reshape(chunks, 64)
    out = src.copy()
    for s in range(STRIPES):
        for k in range(slots):
            k2 = order(k, slots)
            dst = (k * STRIPES + s) * GROUP
            frm = (k2 * STRIPES + s) * GROUP
            out[dst:dst + GROUP] = src[frm:frm + GROUP]
    return out.ravel()


orders = {"default (top half of the image first)": lambda k, m: k,
          "reversed (bottom half first)": lambda k, m: m - 1 - k,
          "bit-reversed (every generation samples the whole image)": lambda k, m: bitrev(k, (m - 1).bit_length()),
          "odd group slots first": lambda k, m: (2 * k + 1) % m if k < m // 2 else (2 * (k - m // 2)) % m,
          "middle out": lambda k, m: (m // 2 + (k + 1) // 2 * (1 if k % 2 else -1)) % m}
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
    "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0),
        "primary 2048x1024": raygen.primary_rays(eye, d, up, fov, 2048, 1024, 0.0, 5000.0)}
# other views of the same scene: from the far end, looking up, looking down, from a corner
if a.cameras:
    c = 0.5 * (lo + hi); e = np.asarray(eye, np.float32)
    for label, (e2, d2) in {"from the other end": (2 * c - e + np.array([0, 2 * (e[1] - c[1]), 0], np.float32), -np.asarray(d, np.float32)),
                            "looking up": (e, np.asarray(d, np.float32) + np.array([0, 0.6, 0], np.float32)),
                            "looking down": (e, np.asarray(d, np.float32) + np.array([0, -0.6, 0], np.float32)),
                            "across": (c + np.array([0, 0, 0.4 * (hi[2] - lo[2])], np.float32),
                                np.array([0.3, -0.1, -1], np.float32))}.items():
        sets["primary, " + label] = raygen.primary_rays(tuple(float(x) for x in e2), tuple(float(x) for x in d2), up, fov, 1024, 1024, 0.0,
            5000.0)
    sets["primary 2048x2048"] = raygen.primary_rays(eye, d, up, fov, 2048, 2048, 0.0, 5000.0)
v = names.index("top-userperm")
for sname, rays in sets.items():
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    base = timed(names.index("top"), rd, hd, n)
    ref = abi.from_device(hd, F.HIT1).tobytes()
    print(f"{sname}: default kernel {base:.4f} ms = {n / base / 1e3:.0f} Mrays/s")
    for label, order in orders.items():
        perm = group_perm(n, order)
        assert np.array_equal(np.sort(perm), np.arange(n))
        pd = torch.from_numpy(perm).cuda()
        abi.lib().rodent_hip_debug_set_perm(0, pd.data_ptr())
        ms = timed(v, rd, hd, n)
        same = abi.from_device(hd, F.HIT1).tobytes() == ref
        print(f"   {label:60s} {ms:.4f} ms  {n / ms / 1e3:7.0f} Mrays/s  identical {same}", flush=True)
    abi.lib().rodent_hip_debug_set_perm(0, None)
