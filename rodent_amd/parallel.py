"""Multi-GPU partitioning: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  The path shards without any data-path collective:
the BVH / scene is replicated, rays or image rows are partitioned, and ONE gather to the root collects the results.

  traversal : rank r gets its own ray batch (sub-pixel sample r of N, or a contiguous ray range)
  frames    : rank r renders the interleaved row tiles row_tiles(height, r, N) (rodent_hip_render_tiles; 16-row tiles dealt round
              robin: contiguous bands of the atrium frame differ by 27 % in cost, tile shares by 1 %, profiles/r04_band_costs.txt)
              or the row band row_band(height, r, N); seeds depend on absolute (sample, iter, x, y) only
              (src/render/renderer.impala:28-33), so any partition reproduces the frame

The gather is a grouped send / receive (what ncclGather is made of): the root posts one receive per peer straight into that
peer's rows of ITS film (or its range of the Hit1 array), every peer posts one send of its own part -- each byte crosses one
xGMI link once, nothing is padded, nobody but the root receives anything (12.4 MB per peer at 3840 x 2160).  The C++ hosts do
the same with ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd (rodent_amd/host/multi_gpu.h).
"""
from __future__ import annotations

import numpy as np


def row_band(height: int, rank: int, world: int):
    """Contiguous row band [y0, y1) of rank `rank`; bands differ by at most one row (2160 / 8 = 270 each)."""
    base, extra = divmod(height, world)
    y0 = rank * base + min(rank, extra)
    return y0, y0 + base + (1 if rank < extra else 0)


TILE_ROWS = 16          # host/partition.h kTileRows


def row_tiles(height: int, rank: int, world: int, tile_rows: int = TILE_ROWS):
    """Interleaved row tiles of rank `rank`: [(y0, y1), ...] = tiles rank, rank + world, ... of `tile_rows` rows each, top to bottom
    (the film's last tile may be shorter).  host/partition.h for_each_tile; what rodent_hip_render_tiles(dev, ..., tile_rows, rank, world)
    renders."""
    return [(t * tile_rows, min((t + 1) * tile_rows, height)) for t in range(rank, (height + tile_rows - 1) // tile_rows, world)]


def ray_range(num_rays: int, rank: int, world: int):
    """Contiguous ray range [a, b): keeps coherent primary rays coherent."""
    base, extra = divmod(num_rays, world)
    a = rank * base + min(rank, extra)
    return a, a + base + (1 if rank < extra else 0)


def _active(dist):
    return dist is not None and dist.is_initialized() and dist.get_world_size() > 1


def gather_parts_to_root(full, part_of, dist, root: int = 0):
    """ONE gather to `root`, in place: `full` is a tensor of the whole result on every rank (only the rank's own part
    `full[part_of(rank)]` holds data); afterwards `full` is complete on the root.  `part_of(r)` -> slice of dim 0, or a list of
    slices (interleaved tiles).  Grouped point-to-point: root receives every peer's parts into their places, peers send theirs."""
    if not _active(dist):
        return full
    rank, world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == "gloo" and full.is_cuda:
        # gloo moves host memory only (CPU tests; bench.py's shared-GPU test mode): bounce the parts through the host
        host = gather_parts_to_root(full.cpu(), part_of, dist, root)
        if host is None:
            return None
        full.copy_(host)
        return full
    def parts(r):
        p = part_of(r)
        return [q for q in (p if isinstance(p, (list, tuple)) else [p]) if q.stop > q.start]
    ops = []
    if rank == root:
        for r in range(world):
            if r != root:
                ops += [dist.P2POp(dist.irecv, full[q], r) for q in parts(r)]
    else:
        ops += [dist.P2POp(dist.isend, full[q], root) for q in parts(rank)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full if rank == root else None


def gather_film_to_root(film, dist=None, root: int = 0, tile_rows: int = 0):
    """The ranks' shares -> the frame on the root.  `film`: [height, width, 3] float32 tensor on the device the collective runs on, the
    rank's share rendered into it (parallel.device_film is such a tensor: a view of the library's film): its band
    row_band(height, rank, world), or with tile_rows > 0 its interleaved tiles row_tiles(height, rank, world, tile_rows).
    Returns the complete film on the root (the same tensor, completed in place), None on the other ranks."""
    if not _active(dist):
        return film
    height, world = film.shape[0], dist.get_world_size()
    if tile_rows > 0:
        return gather_parts_to_root(film, lambda r: [slice(a, b) for a, b in row_tiles(height, r, world, tile_rows)], dist, root)
    return gather_parts_to_root(film, lambda r: slice(*row_band(height, r, world)), dist, root)


def gather_hits_to_root(hits_bytes, num_rays: int, dist=None, root: int = 0):
    """Hit1 ranges -> the whole Hit1 array on the root.  `hits_bytes`: uint8 tensor of 16 x num_rays bytes on every rank, the
    rank's own ray_range filled in.  Completed in place on the root; None on the other ranks."""
    if not _active(dist):
        return hits_bytes
    world = dist.get_world_size()

    def part(r):
        a, b = ray_range(num_rays, r, world)
        return slice(a * 16, b * 16)
    return gather_parts_to_root(hits_bytes, part, dist, root)


def device_film(dev: int):
    """The renderer's DEVICE film (rodent_get_film_data, interface.cpp:565-581) as a torch tensor [height, width, 3] that
    aliases the library's memory (no copy): what gather_film_to_root hands to RCCL."""
    import ctypes as C
    import torch
    from . import render
    l = render.stage_lib()
    ptr, w, h = C.c_void_p(), C.c_int32(), C.c_int32()
    l.rodent_get_film_data(dev, C.byref(ptr), C.byref(w), C.byref(h))

    class _Alias:                                   # zero-copy view of foreign device memory
        __cuda_array_interface__ = {"shape": (h.value, w.value, 3), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
    return torch.as_tensor(_Alias(), device=f"cuda:{dev}")


def gather_film(band: np.ndarray, height: int, dist=None, device="cpu", root: int = 0):
    """Host-array form (the CPU tests run it over gloo): this rank's band [rows_r, width, 3] float32 -> the assembled
    [height, width, 3] array on the root, None elsewhere."""
    import torch
    if not _active(dist):
        return band
    y0, y1 = row_band(height, dist.get_rank(), dist.get_world_size())
    full = torch.zeros((height,) + tuple(band.shape[1:]), dtype=torch.float32, device=device)
    full[y0:y1] = torch.from_numpy(np.ascontiguousarray(band)).to(device)
    out = gather_film_to_root(full, dist, root)
    return None if out is None else out.cpu().numpy()


def gather_hits_device(hits_dev, num_rays: int, dist, dev: int, root: int = 0):
    """bench.py (strong scaling): the rank's device Hit1 range -> host Hit1 array of all rays on the root (the D2H copy
    follows the collective, outside the timed region like the reference's, bench_traversal.cpp:337-339)."""
    import torch
    from . import formats as F
    if not _active(dist):
        return hits_dev.cpu().numpy().view(F.HIT1).copy()
    a, b = ray_range(num_rays, dist.get_rank(), dist.get_world_size())
    full = torch.zeros(num_rays * 16, dtype=torch.uint8, device=f"cuda:{dev}")
    full[a * 16: b * 16] = hits_dev[: (b - a) * 16]
    out = gather_hits_to_root(full, num_rays, dist, root)
    return None if out is None else out.cpu().numpy().view(F.HIT1).copy()


def gather_hits(hits: np.ndarray, num_rays: int, dist=None, device="cpu", root: int = 0):
    """Host-array form of gather_hits_to_root (CPU tests over gloo): the root gets the whole array, the others None."""
    import torch
    from . import formats as F
    if not _active(dist):
        return hits
    a, b = ray_range(num_rays, dist.get_rank(), dist.get_world_size())
    full = torch.zeros(num_rays * 16, dtype=torch.uint8, device=device)
    full[a * 16: b * 16] = torch.from_numpy(np.ascontiguousarray(hits).view(np.uint8).reshape(-1).copy()).to(device)
    out = gather_hits_to_root(full, num_rays, dist, root)
    return None if out is None else out.cpu().numpy().view(F.HIT1).copy()
