"""ctypes binding of librodent_hip.so (the C ABI declared in include/rodent_traversal.h).

This is the Python-side mirror of what the reference's bench_traversal.cpp does
with the AnyDSL-generated traversal.h: hand device pointers to the entry points.
Device memory comes from torch (plumbing only); every entry point receives raw
pointers.  There is NO CPU fallback: if the shared library is missing or no GPU is
visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import formats as F

# RODENT_HIP_LAB=1 selects the lab build (librodent_hip_lab.so: the product kernels plus every measured-and-lost variant
# and the instrumented builds; made by `RODENT_HIP_LAB=1 python -m rodent_amd.build`).  Default: the product library.
LAB = os.environ.get("RODENT_HIP_LAB", "0") not in ("", "0")
LIB_PATH = Path(__file__).resolve().parent / "lib" / ("librodent_hip_lab.so" if LAB else "librodent_hip.so")
# lab: another build of the library (A/B runs of compiler options, scripts/flags_experiment.sh)
if os.environ.get("RODENT_HIP_LIB"):
    LIB_PATH = Path(os.environ["RODENT_HIP_LIB"]).resolve()

SYNC_ENTRY_POINTS = [
    "amdgpu_intersect_single_ray1_bvh2_tri1", "amdgpu_occluded_single_ray1_bvh2_tri1",
    "hip_intersect_single_ray1_bvh4_tri4", "hip_occluded_single_ray1_bvh4_tri4",
    "hip_intersect_single_ray1_bvh8_tri4", "hip_occluded_single_ray1_bvh8_tri4",
]
ASYNC_ENTRY_POINTS = ["hip_traverse_bvh2_tri1_async", "hip_traverse_bvh4_tri4_async", "hip_traverse_bvh8_tri4_async"]
EXPORTS = SYNC_ENTRY_POINTS + ASYNC_ENTRY_POINTS + [
    "rodent_hip_check_errors", "rodent_hip_get_kernel_time", "rodent_hip_device_count", "rodent_hip_num_variants",
        "rodent_hip_variant_name",
    "rodent_hip_kernel_name", "rodent_hip_version", "rodent_hip_source_digest", "rodent_hip_is_lab_build", "rodent_hip_phased_min_rays",
        "rodent_hip_top_min_rays", "rodent_hip_ray_kind_hint", "rodent_hip_ray_grid", "rodent_hip_schedule_history",
        "rodent_hip_read_stats", "rodent_hip_read_trace", "rodent_hip_debug_set_perm",
]
BLOCK_OF_WIDTH = {2: F.BVH2_TRI1, 4: F.BVH4_TRI4, 8: F.BVH8_TRI4}

_lib = None


class MissingExtension(RuntimeError):
    pass


def lib():
    """Loads librodent_hip.so (built in-tree by rodent_amd.build); raises if absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MissingExtension(f"{LIB_PATH} not built: run `python -m rodent_amd.build` (needs hipcc)")
        l = C.CDLL(str(LIB_PATH))
        vp, i32 = C.c_void_p, C.c_int32
        for name in SYNC_ENTRY_POINTS:
            fn = getattr(l, name); fn.restype = None; fn.argtypes = [i32, vp, vp, vp, vp, i32]
        for name in ASYNC_ENTRY_POINTS:
            fn = getattr(l, name); fn.restype = None; fn.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, vp]
        l.rodent_hip_check_errors.restype = i32; l.rodent_hip_check_errors.argtypes = [i32, vp]
        l.rodent_hip_is_lab_build.restype = i32; l.rodent_hip_is_lab_build.argtypes = []
        l.rodent_hip_phased_min_rays.restype = None; l.rodent_hip_phased_min_rays.argtypes = [i32]
        l.rodent_hip_top_min_rays.restype = None; l.rodent_hip_top_min_rays.argtypes = [i32]
        l.rodent_hip_get_kernel_time.restype = C.c_uint64; l.rodent_hip_get_kernel_time.argtypes = []
        l.rodent_hip_ray_kind_hint.restype = None; l.rodent_hip_ray_kind_hint.argtypes = [i32]
        l.rodent_hip_ray_grid.restype = None; l.rodent_hip_ray_grid.argtypes = [i32]
        l.rodent_hip_schedule_history.restype = None; l.rodent_hip_schedule_history.argtypes = [i32]
        l.rodent_hip_device_count.restype = i32; l.rodent_hip_device_count.argtypes = []
        l.rodent_hip_num_variants.restype = i32; l.rodent_hip_num_variants.argtypes = [i32]
        l.rodent_hip_variant_name.restype = C.c_char_p; l.rodent_hip_variant_name.argtypes = [i32, i32]
        l.rodent_hip_kernel_name.restype = C.c_char_p; l.rodent_hip_kernel_name.argtypes = [i32, i32, i32]
        l.rodent_hip_version.restype = C.c_char_p; l.rodent_hip_version.argtypes = []
        l.rodent_hip_source_digest.restype = C.c_char_p; l.rodent_hip_source_digest.argtypes = []
        l.rodent_hip_read_trace.restype = None; l.rodent_hip_read_trace.argtypes = [i32, C.c_void_p]
        l.rodent_hip_debug_set_perm.restype = None; l.rodent_hip_debug_set_perm.argtypes = [i32, C.c_void_p]
        l.rodent_hip_read_stats.restype = None; l.rodent_hip_read_stats.argtypes = [i32, C.POINTER(C.c_uint64)]
        _lib = l
    return _lib


def built_from_these_sources() -> bool:
    """Whether the loaded library was compiled from the sources in this tree (its compiled-in digest against build.source_digest())."""
    from . import build
    return lib().rodent_hip_source_digest().decode() == build.source_digest()


def variants(width):
    l = lib()
    return [l.rodent_hip_variant_name(width, i).decode() for i in range(l.rodent_hip_num_variants(width))]


# Mappings that do NOT keep the reference's per-ray visit order (lab build only: work stealing inside the wave, measured and lost): last-bit
# ties in t resolve to another triangle.  Every shipped mapping reproduces the reference kernel bit for bit (tests/). (round 6: deferred
# leaves / triangle turns with deferral visit a superset of the nodes)
ORDER_CHANGING = ("steal", "defer-", "stats-defer", "turns-p")


def order_preserving_variants(width):
    """Indices of the mappings whose hit records equal the reference kernel's bit for bit."""
    return [i for i, n in enumerate(variants(width)) if not n.startswith(ORDER_CHANGING)]


def kernel_name(width, variant, any_hit=False):
    return lib().rodent_hip_kernel_name(width, variant, int(any_hit)).decode()


# ---- device buffers (torch is plumbing: allocation + copies) ---------------------------------

def to_device(arr: np.ndarray, dev: int = 0):
    """Structured numpy array -> torch uint8 CUDA tensor holding the same bytes."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("rodent_amd: no GPU visible (torch.cuda.is_available() is False)")
    a = np.ascontiguousarray(arr)
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy() if a.size else np.zeros(0, np.uint8))
    return t.to(f"cuda:{dev}")


def from_device(t, dtype: np.dtype) -> np.ndarray:
    return t.cpu().numpy().view(dtype).copy()


class DeviceBvh:
    """A BVH resident in HBM: nodes + tris of one layout (2 = Node2/Tri1, 4 = Node4/Tri4, 8 = Node8/Tri4)."""

    def __init__(self, width, nodes, tris, dev=0):
        assert width in (2, 4, 8)
        self.width, self.dev = width, dev
        self.num_nodes, self.num_tris = len(nodes), len(tris)
        self.nodes = to_device(nodes, dev)
        self.tris = to_device(tris, dev)

    @classmethod
    def load(cls, path, width, dev=0):
        nodes, tris = F.read_bvh(path, BLOCK_OF_WIDTH[width])
        return cls(width, nodes, tris, dev)


def traverse_async(bvh: DeviceBvh, rays_dev, hits_dev, num_rays, any_hit=False, variant=0, stream=None):
    """Enqueues one traversal pass on `stream` (torch stream or None = torch's current stream)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(bvh.dev)
    fn = getattr(lib(), {2: "hip_traverse_bvh2_tri1_async", 4: "hip_traverse_bvh4_tri4_async",
        8: "hip_traverse_bvh8_tri4_async"}[bvh.width])
    fn(bvh.dev, bvh.nodes.data_ptr(), bvh.tris.data_ptr(), rays_dev.data_ptr(), hits_dev.data_ptr(),
       int(num_rays), int(any_hit), int(variant), C.c_void_p(stream.cuda_stream))


def traverse(bvh: DeviceBvh, rays: np.ndarray, any_hit=False, variant=None) -> np.ndarray:
    """Synchronous convenience wrapper: host rays in, host Hit1 array out.

    With variant=None it goes through the reference-named synchronous entry points
    (amdgpu_intersect_single_ray1_bvh2_tri1 & co)."""
    import torch
    n = len(rays)
    rays_dev = to_device(rays, bvh.dev)
    hits_dev = torch.zeros(max(n, 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{bvh.dev}")
    if variant is None:
        torch.cuda.synchronize(bvh.dev)
        name = {(2, False): "amdgpu_intersect_single_ray1_bvh2_tri1", (2, True): "amdgpu_occluded_single_ray1_bvh2_tri1",
                (4, False): "hip_intersect_single_ray1_bvh4_tri4", (4, True): "hip_occluded_single_ray1_bvh4_tri4",
                (8, False): "hip_intersect_single_ray1_bvh8_tri4",
                    (8, True): "hip_occluded_single_ray1_bvh8_tri4"}[(bvh.width, bool(any_hit))]
        getattr(lib(), name)(bvh.dev, bvh.nodes.data_ptr(), bvh.tris.data_ptr(), rays_dev.data_ptr(), hits_dev.data_ptr(), n)
    else:
        traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant)
        torch.cuda.synchronize(bvh.dev)
        check_errors(bvh.dev)
    return from_device(hits_dev, F.HIT1)[:n]


def schedule_history(enable: bool):
    """rodent_hip_schedule_history: the default BVH2 mapping traces its chunks longest-first by the per-chunk cost the previous
    launch of the same size recorded (off by default; per (device, stream); hit records do not depend on it)."""
    lib().rodent_hip_schedule_history(int(bool(enable)))


def top_min_rays(rays: int):
    """rodent_hip_top_min_rays: launches of fewer rays take the one-chunk kernel instead of the persistent LDS-image kernel
    (< 0: the shipped threshold, 0: every launch through the LDS-image kernel)."""
    lib().rodent_hip_top_min_rays(int(rays))


def ray_kind_hint(enable: bool):
    """rodent_hip_ray_kind_hint: may the default BVH2 mapping remember that a ray list (pointer, count) was incoherent and trace it with
    the refill kernel from its second launch on (default: no -- kernel selection is stateless; hit records do not depend on it)."""
    lib().rodent_hip_ray_kind_hint(int(bool(enable)))


def ray_grid(width: int = -1):
    """rodent_hip_ray_grid: -1 = the default BVH2 kernel recognises camera rays in image order and traces them as 8 x 8-pixel tiles
    (default), 0 = never, > 0 = that image width on trust (hit records do not depend on it)."""
    lib().rodent_hip_ray_grid(int(width))


def check_errors(dev=0, stream=None):
    """The asynchronous entry points report a traversal-stack overflow (more than the reference's 64 entries,
    stack.impala:53) through a device-side flag: this waits for `stream`, reads and clears it, and raises."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(dev)
    if lib().rodent_hip_check_errors(dev, C.c_void_p(stream.cuda_stream)):
        raise RuntimeError("rodent_hip: traversal stack overflow (more than 64 entries)")


def read_stats(dev=0):
    """Phase counters of the instrumented "stats-*" variants (reads and clears)."""
    buf = (C.c_uint64 * 8)()
    lib().rodent_hip_read_stats(dev, buf)
    return list(buf)


def read_trace(dev=0, arm_only=False):
    """Per-wave timeline of the instrumented variants: array [16384, 4] of uint64 (see the header)."""
    if arm_only:
        lib().rodent_hip_read_trace(dev, None)
        return None
    buf = np.zeros((16384, 4), np.uint64)
    lib().rodent_hip_read_trace(dev, buf.ctypes.data_as(C.c_void_p))
    return buf
