"""Timing and rank bookkeeping of bench.py: the timed region (W untimed + K timed launches between barriers), reductions over ranks."""
from __future__ import annotations

import json
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def time_passes(abi, torch, bvh, rays_dev, hits_dev, n, variant, steps, warmup, dist, any_hit=False):
    """W untimed + K timed launches, back to back on one stream.  Returns (wall seconds for the K steps [max over ranks is taken by the
    caller], average launch duration in ms from ONE pair of HIP events around the timed region on the launch stream, ... and, from a second,
    untimed pass with an event pair around every single launch, the median and minimum of those).  Until round 4 the timed region itself
    carried an event pair per step: two marker packets between every two kernels cost 17 us per 0.18 ms step
    (profiles/r05_host_call_costs.txt: 174.9 us per launch back to back against 192.3 with them) -- time the benchmark spent measuring
    itself."""
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    first.record(stream)
    for i in range(steps):
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
    last.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    region_ms = first.elapsed_time(last) / max(1, steps)
    # per-launch spread (not part of the timed region): every launch between its own two events, which adds the dispatch latency the
    # back-to-back region hides
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
        starts[i].record(stream)
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
        ends[i].record(stream)
    torch.cuda.synchronize()
    single = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    return wall, float(region_ms), float(np.median(single)), float(np.min(single))


def _collective_device(dist, dev):
    return "cpu" if dist.get_backend() == "gloo" else f"cuda:{dev}"


def max_over_ranks(torch, dist, dev, values):
    if dist is None:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device(dist, dev))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def gather_scalars(torch, dist, dev, values):
    """[[values of rank 0], [values of rank 1], ...] on every rank (bookkeeping, after the timed regions)."""
    if dist is None:
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device(dist, dev))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in g] for g in out]
