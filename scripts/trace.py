#!/usr/bin/env python
"""Per-wave timeline of the instrumented sched kernels: lifetimes, per-CU placement, occupancy over time."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
sets = {"primary": raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0),
        "random": raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)}
names = abi.variants(2)
abi.read_trace(arm_only=True)
want = sys.argv[1:] or ["stats-sched", "stats-sched-p16"]
for name in want:
    v = names.index(name)
    for k, rays in sets.items():
        n = len(rays)
        rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
        for _ in range(2):
            abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize(); abi.read_trace()
        abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize()
        tr = abi.read_trace(); abi.read_stats()
        tr = tr[tr[:, 1] > 0]
        t0 = tr[:, 0].min(); start = (tr[:, 0] - t0) / 100.0; end = (tr[:, 1] - t0) / 100.0   # us (100 MHz)
        hw = tr[:, 2] & 0xFFFFFFFF; xcc = (tr[:, 2] >> 32) & 0xF
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        rays_w = tr[:, 3] >> 32; outer = tr[:, 3] & 0xFFFFFFFF
        if name == "trace-fast":                       # word 3 = node iterations << 32 | leaf iterations
            itn, itl = rays_w.astype(float), outer.astype(float)
            life_ = end - start
            print(f"   iterations/wave: node mean {itn.mean():.1f} p50 {np.median(itn):.0f} p99 {np.percentile(itn, 99):.0f} max "
                f"{itn.max():.0f}; leaf mean {itl.mean():.1f} max {itl.max():.0f}")
            A = np.stack([itn, itl, np.ones_like(itn)], 1); coef = np.linalg.lstsq(A, life_, rcond=None)[0]
            print(f"   life ~ {coef[0]:.3f} us/node-iter + {coef[1]:.3f} us/leaf-iter + {coef[2]:.2f} us")
            early = start < np.percentile(start, 25); late = start > np.percentile(start, 90)
            for lab, m in (("first 25% started", early), ("last 10% started", late)):
                c = np.linalg.lstsq(A[m], life_[m], rcond=None)[0]
                print(f"   {lab}: life mean {life_[m].mean():.1f} us, node-iter mean {itn[m].mean():.1f}, fit "
                    f"{c[0]:.3f}/{c[1]:.3f}/{c[2]:.2f}")
            order = np.argsort(-end)[:8]
            print("   last waves to end: "
                + "; ".join(f"blk {int(np.flatnonzero(tr[:, 0] == tr[o, 0])[0])} start {start[o]:.0f} life {life_[o]:.0f} "
                f"n{itn[o]:.0f}/l{itl[o]:.0f}" for o in order))
        life = end - start
        uniq, cnt = np.unique(cuid, return_counts=True)
        ts = np.linspace(0, end.max(), 21)[1:-1]
        occ = [(int(((start <= t) & (end > t)).sum())) for t in ts]
        print(f"{name} {k}: waves {len(tr)} span {end.max():.1f}us  life mean {life.mean():.1f} p50 {np.median(life):.1f} max "
            f"{life.max():.1f}us | "
              f"start max {start.max():.1f}us | CUs used {len(uniq)} waves/CU min {cnt.min()} max {cnt.max()} | rays/wave mean "
                  f"{rays_w.mean():.0f} max {rays_w.max()} | outer mean {outer.mean():.0f}")
        print("   active waves at 5%..95% of span:", occ)
        # per-CU busy time
        busy = np.array([end[cuid == u].max() for u in uniq])
        print(f"   per-CU last-wave-end: min {busy.min():.1f} p50 {np.median(busy):.1f} max {busy.max():.1f} us; per XCC waves: "
            f"{np.bincount(xcc.astype(int), minlength=8).tolist()}")
