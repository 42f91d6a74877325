#!/usr/bin/env python
"""Model of VERDICT r5 item 3 (CPU only: the oracle's per-ray step counts, no GPU): what a STATELESS in-launch cost probe could be worth to
the default
mapping's 1 Mi-ray launch.  The launch: 64 stripes x 128 resident waves; stripe s owns the 32-chunk groups g = s mod 64 (8 groups = 256
chunks at 1 Mi rays);
a wave's first chunk is its rank in the stripe (groups 0..3 of the stripe), the counter hands out the rest in order.  A chunk costs
max(steps of its rays) wave
iterations; list scheduling per stripe at a fixed iteration time (the replay that matched the hardware to a few per cent in round 2,
LAB_NOTES 3.1.1).
Orders compared:
  default        groups in list order
  history        every stripe's chunks longest first by their TRUE cost (what rodent_hip_schedule_history reaches on identical launches)
  probe-1ray     the proposal: one ray per group (ray 0 of the group's middle chunk) traced first; groups in descending order of its step
  count; the probing wave
                 starts its own first chunk late by the longest probe ray
  probe-8rays    eight rays per group (ray 0 of every fourth chunk), otherwise the same
  group-oracle   groups in descending order of their TRUE maximum chunk cost (the ceiling of any group-granular order)
  ... known after K  the buildable form of the probe: see span_capped
usage: python scripts/model_probe_order.py [scene ...]"""
import sys
from pathlib import Path
import heapq
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import formats as F, raygen, scenes
from oracle import binding as O

STRIPES, GROUP, WAVES = 64, 32, 128


def tiles(w, h):
    ty, tx = np.meshgrid(np.arange(h // 8), np.arange(w // 8), indexing="ij")
    jy, jx = np.arange(64) // 8, np.arange(64) % 8
    return ((ty.ravel()[:, None] * 8 + jy[None, :]) * w + tx.ravel()[:, None] * 8 + jx[None, :]).ravel()


def span(cost, order_of_stripe, late=None):
    """Longest stripe: WAVES slots per stripe take the stripe's chunks in the given order (the first WAVES of them at t = 0)."""
    worst = 0.0
    for s in range(STRIPES):
        chunks = order_of_stripe(s)
        slots = [0.0] * WAVES
        if late is not None:
            slots[0] = late[s]
        heapq.heapify(slots)
        end = 0.0
        for c in chunks:
            t = heapq.heappop(slots) + cost[c]
            end = max(end, t)
            heapq.heappush(slots, t)
        worst = max(worst, end)
    return worst


def span_capped(cost, groups, est_ray, cap):
    """The buildable form: the first generation starts at t = 0 on the default groups; one wave per stripe traces the probe rays for at most
    `cap` iterations (estimate = min(steps, cap)) and starts its own first chunk `cap` late; a draw before t = cap takes the next chunk in
    default order, a draw after it the next chunk of the undrawn group with the highest estimate."""
    worst = 0.0
    for s in range(STRIPES):
        gs = np.arange(s, groups, STRIPES)
        est = np.minimum(est_ray[gs], cap)
        default_order = [int(c) for g in gs for c in range(g * GROUP, (g + 1) * GROUP)]
        probe_order = [int(c) for g in gs[np.argsort(-est, kind="stable")] for c in range(g * GROUP, (g + 1) * GROUP)]
        drawn = set()
        slots = [0.0] * WAVES
        slots[0] = float(cap)
        heapq.heapify(slots)
        di = pi = 0
        end = 0.0
        for _ in range(len(default_order)):
            t0 = heapq.heappop(slots)
            if t0 < cap:
                while default_order[di] in drawn: di += 1
                c = default_order[di]
            else:
                while probe_order[pi] in drawn: pi += 1
                c = probe_order[pi]
            drawn.add(c)
            t = t0 + cost[c]
            end = max(end, t)
            heapq.heappush(slots, t)
        worst = max(worst, end)
    return worst


def study(name, steps):
    n = len(steps)
    chunks = n // 64
    cost = steps.reshape(chunks, 64).max(axis=1).astype(np.float64)
    groups = chunks // GROUP
    stripe_groups = lambda s: np.arange(s, groups, STRIPES)
    chunks_of = lambda gs: (gs[:, None] * GROUP + np.arange(GROUP)[None, :]).ravel()
    default = span(cost, lambda s: chunks_of(stripe_groups(s)))
    history = span(cost, lambda s: sorted(chunks_of(stripe_groups(s)), key=lambda c: -cost[c]))
    first_ray = steps.reshape(chunks, 64)[:, 0].astype(np.float64)

    def by_estimate(est):
        return lambda s: chunks_of(stripe_groups(s)[np.argsort(-est[stripe_groups(s)], kind="stable")])
    est1 = first_ray[np.arange(groups) * GROUP + GROUP // 2]
    est8 = first_ray.reshape(groups, GROUP)[:, ::4].max(axis=1)
    true_max = cost.reshape(groups, GROUP).max(axis=1)
    late1 = [float(est1[stripe_groups(s)].max()) for s in range(STRIPES)]
    late8 = [float(est8[stripe_groups(s)].max()) for s in range(STRIPES)]
    rows = [("default", default), ("history (true chunk costs, stateful)", history), ("probe-1ray", span(cost, by_estimate(est1), late1)),
            ("probe-8rays", span(cost, by_estimate(est8), late8)),
                ("group-oracle (ceiling of group-granular orders)", span(cost, by_estimate(true_max)))]
    for cap in (24, 32, 48, 64, 96):
        rows.append((f"probe-1ray, known after {cap} iterations (buildable)", span_capped(cost, groups, est1, cap)))
    rows.append(("true group maximum, known after 48 iterations", span_capped(cost, groups, true_max, 48)))
    rows.append(("true group maximum, known at t = 0, first generation on default groups", span_capped(cost, groups, true_max + 1e6, 0)))
    corr1 = np.corrcoef(est1, true_max)[0, 1]
    print(f"{name}: {chunks} chunks, mean chunk {cost.mean():.1f} iterations, longest {cost.max():.0f}; work per slot "
        f"{cost.sum() / (STRIPES * WAVES):.1f}; "
          f"corr(probe ray, group's longest chunk) {corr1:.2f}")
    for label, v in rows:
        print(f"    {label:52s} span {v:7.1f} iterations   {default / v:5.3f} x default")


for scene in (sys.argv[1:] or ["atrium"]):
    path = scenes.scene_bvh(scene)
    nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
    eye, d, up, fov = scenes.CAMERAS[scene.split("/")[0]]
    prim = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX)
    lo, hi = raygen.scene_bounds2(nodes)
    rnd = raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, scenes.RANDOM_TMAX)
    study(f"{scene} camera rays as 8 x 8 tiles", O.ray_steps(nodes, tris, prim).sum(axis=1)[tiles(1024, 1024)])
    study(f"{scene} random segments (whole chunks)", O.ray_steps(nodes, tris, rnd).sum(axis=1))
