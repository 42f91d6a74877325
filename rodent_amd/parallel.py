"""Multi-GPU partitioning: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  The path shards without any data-path collective:
the BVH / scene is replicated, rays or image rows are partitioned, and ONE gather collects the results.

  traversal : rank r gets its own ray batch (sub-pixel sample r of N, or a contiguous ray range)
  frames    : rank r renders the row band row_band(height, r, N); seeds depend on absolute
              (sample, iter, x, y) only (src/render/renderer.impala:28-33), so bands reproduce the frame
"""
from __future__ import annotations

import numpy as np


def row_band(height: int, rank: int, world: int):
    """Contiguous row band [y0, y1) of rank `rank`; bands differ by at most one row (2160 / 8 = 270 each)."""
    base, extra = divmod(height, world)
    y0 = rank * base + min(rank, extra)
    return y0, y0 + base + (1 if rank < extra else 0)


def ray_range(num_rays: int, rank: int, world: int):
    """Contiguous ray range [a, b): keeps coherent primary rays coherent."""
    base, extra = divmod(num_rays, world)
    a = rank * base + min(rank, extra)
    return a, a + base + (1 if rank < extra else 0)


def gather_film(band: np.ndarray, height: int, dist=None, device="cpu"):
    """One gather of the per-rank row bands (each [rows_r, width, 3] float32) into the full film on every
    rank (all_gather on equal-size padded slabs: RCCL has no gather primitive; the payload at 3840x2160 is
    12.4 MB per GPU).  Returns the assembled [height, width, 3] array."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return band
    world, rank = dist.get_world_size(), dist.get_rank()
    width = band.shape[1]
    max_rows = max(row_band(height, r, world)[1] - row_band(height, r, world)[0] for r in range(world))
    slab = torch.zeros((max_rows, width, 3), dtype=torch.float32, device=device)
    slab[: band.shape[0]] = torch.from_numpy(np.ascontiguousarray(band)).to(device)
    out = [torch.zeros_like(slab) for _ in range(world)]
    dist.all_gather(out, slab)
    film = np.zeros((height, width, 3), np.float32)
    for r in range(world):
        y0, y1 = row_band(height, r, world)
        film[y0:y1] = out[r][: y1 - y0].cpu().numpy()
    return film


def gather_hits(hits: np.ndarray, num_rays: int, dist=None, device="cpu"):
    """One gather of per-rank Hit1 ranges (16 B/ray) into the full array on every rank."""
    import torch
    from . import formats as F
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return hits
    world = dist.get_world_size()
    max_n = max(ray_range(num_rays, r, world)[1] - ray_range(num_rays, r, world)[0] for r in range(world))
    slab = torch.zeros(max_n * 16, dtype=torch.uint8, device=device)
    raw = torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1).copy()).to(device)
    slab[: raw.numel()] = raw
    out = [torch.zeros_like(slab) for _ in range(world)]
    dist.all_gather(out, slab)
    full = np.zeros(num_rays, F.HIT1)
    for r in range(world):
        a, b = ray_range(num_rays, r, world)
        full[a:b] = out[r][: (b - a) * 16].cpu().numpy().view(F.HIT1)
    return full
