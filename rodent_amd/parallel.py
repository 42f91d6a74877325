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


def _gather_slabs(slab, counts, dist):
    """ONE all_gather of equal-size slabs (RCCL has no gatherv: every rank pads to the largest share); returns the list of
    per-rank tensors cut back to their true length.  `slab` lives where the collective runs (GPU for RCCL, CPU for gloo)."""
    import torch
    world = dist.get_world_size()
    longest = max(counts)
    padded = slab if slab.shape[0] == longest else torch.cat([slab, torch.zeros((longest - slab.shape[0],) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)])
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded.contiguous())
    return [out[r][: counts[r]] for r in range(world)]


def gather_film_tensor(band, height: int, dist=None):
    """Row bands -> full film, as torch tensors on the device the collective runs on: band [rows_r, width, 3] float32 of
    this rank -> [height, width, 3] on every rank.  One all_gather, no host bounce (12.4 MB per GPU at 3840x2160)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return band
    world = dist.get_world_size()
    rows = [row_band(height, r, world)[1] - row_band(height, r, world)[0] for r in range(world)]
    return torch.cat(_gather_slabs(band, rows, dist), dim=0)


def device_film(dev: int):
    """The renderer's DEVICE film (rodent_get_film_data, interface.cpp:565-581) as a torch tensor [height, width, 3] that
    aliases the library's memory (no copy): what gather_film_tensor hands to RCCL."""
    import ctypes as C
    import torch
    from . import render
    l = render.stage_lib()
    ptr, w, h = C.c_void_p(), C.c_int32(), C.c_int32()
    l.rodent_get_film_data(dev, C.byref(ptr), C.byref(w), C.byref(h))

    class _Alias:                                   # zero-copy view of foreign device memory
        __cuda_array_interface__ = {"shape": (h.value, w.value, 3), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
    return torch.as_tensor(_Alias(), device=f"cuda:{dev}")


def gather_film(band: np.ndarray, height: int, dist=None, device="cpu"):
    """Host-array form of gather_film_tensor (the CPU tests run it over gloo): band [rows_r, width, 3] float32 -> the
    assembled [height, width, 3] array on every rank."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return band
    return gather_film_tensor(torch.from_numpy(np.ascontiguousarray(band)).to(device), height, dist).cpu().numpy()


def gather_hits_tensor(hits_bytes, num_rays: int, dist=None):
    """Hit1 ranges (uint8 tensor, 16 B/ray, this rank's ray_range) -> the whole array in ray order on every rank; one
    all_gather on the device the tensor lives on."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return hits_bytes
    world = dist.get_world_size()
    counts = [(ray_range(num_rays, r, world)[1] - ray_range(num_rays, r, world)[0]) * 16 for r in range(world)]
    return torch.cat(_gather_slabs(hits_bytes[: counts[dist.get_rank()]], counts, dist), dim=0)


def gather_hits_device(hits_dev, num_rays: int, dist, dev: int):
    """bench.py --strong: device Hit1 ranges -> host Hit1 array of all rays (the D2H copy follows the collective, outside
    the timed region like the reference's, bench_traversal.cpp:337-339)."""
    from . import formats as F
    return gather_hits_tensor(hits_dev, num_rays, dist).cpu().numpy().view(F.HIT1).copy()


def gather_hits(hits: np.ndarray, num_rays: int, dist=None, device="cpu"):
    """Host-array form of gather_hits_tensor (CPU tests over gloo)."""
    import torch
    from . import formats as F
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return hits
    raw = torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1).copy()).to(device)
    return gather_hits_tensor(raw, num_rays, dist).cpu().numpy().view(F.HIT1).copy()
