#!/usr/bin/env python
"""Per-chunk timeline of the persistent default kernel (lab variant "trace-top"): when every 64-ray chunk started and ended,
how many iterations it took, which wave took it -- and what a different chunk order could gain: a list-scheduling replay of
the measured chunk durations on the same 8192 wave slots, in ticket order and longest-first.
usage: RODENT_HIP_LAB=1 python scripts/trace_top.py"""
import sys
from pathlib import Path
import heapq
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
v = abi.variants(2).index("trace-top")
abi.read_trace(arm_only=True)


def replay(durations, slots=8192):
    """Greedy list scheduling: every chunk goes to the slot that frees first; returns the makespan."""
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for dur in durations:
        t = heapq.heappop(free) + dur
        end = max(end, t)
        heapq.heappush(free, t)
    return end


for k, rays in sets.items():
    n = len(rays)
    rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize(); abi.read_trace()
    abi.traverse_async(bvh, rd, hd, n, False, v); torch.cuda.synchronize()
    tr = abi.read_trace()
    tr = tr[tr[:, 1] > 0]
    t0 = tr[:, 0].min(); start = (tr[:, 0] - t0) / 100.0; end = (tr[:, 1] - t0) / 100.0; it = tr[:, 2].astype(float)
    wave = (tr[:, 3] >> 32).astype(int); ticket = (tr[:, 3] & 0xFFFFFFFF).astype(int)
    dur = end - start
    second = ticket >= 128          # 512 workgroups x 16 waves / 64 stripes
    print(f"{k}: {len(tr)} chunks, span {end.max():.1f} us; chunk duration mean {dur.mean():.1f} p50 {np.median(dur):.1f} p99 "
        f"{np.percentile(dur, 99):.1f} max {dur.max():.1f} us; "
          f"iterations mean {it.mean():.1f} max {it.max():.0f}; us per iteration mean {(dur / np.maximum(it, 1)).mean():.3f}")
    print(f"   first generation: start max {start[~second].max():.1f}, end mean {end[~second].mean():.1f} max {end[~second].max():.1f}; "
        f"drawn chunks: start mean {start[second].mean():.1f} max {start[second].max():.1f}, end max {end[second].max():.1f}")
    print("   first-generation start times (us): p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f"
        % tuple(np.percentile(start[~second], [1, 10, 50, 90, 99, 100])))
    ts = np.linspace(0, end.max(), 21)[1:-1]
    print("   chunks in flight at 5%..95% of the span:", [int(((start <= t) & (end > t)).sum()) for t in ts])
    last = np.argsort(-end)[:6]
    print("   last chunks to end: " + "; ".join(f"ticket {ticket[o]} start {start[o]:.0f} dur {dur[o]:.0f} it {it[o]:.0f}" for o in last))
    # iteration time under load vs at the end
    for lab, m in (("started before 20 us", start < 20), ("started after 60 us", start > 60)):
        if m.sum():
            print(f"   {lab}: {int(m.sum())} chunks, us/iteration {(dur[m] / np.maximum(it[m], 1)).mean():.3f}, duration mean "
                f"{dur[m].mean():.1f}")
    order = np.argsort(start)
    print(f"   replay of the measured durations on 8192 slots: in start order {replay(dur[order]):.1f} us, longest first "
        f"{replay(np.sort(dur)[::-1]):.1f} us, "
          f"lower bounds: work / slots {dur.sum() / 8192:.1f} us, longest chunk {dur.max():.1f} us")
    # the same with durations rescaled to iterations x the unloaded rate (what the chain of the longest ray costs alone)
    print(f"   critical chain: {it.max():.0f} iterations x {np.percentile(dur / np.maximum(it, 1), 5):.3f} us (fastest 5 % of the chunks) = "
        f"{it.max() * np.percentile(dur / np.maximum(it, 1), 5):.1f} us")
