"""ctypes binding of the CPU parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under rodent_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"

HIT1 = np.dtype([("tri_id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


class OracleStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("inner_nodes", C.c_uint64), ("prim_packets", C.c_uint64),
                ("hits", C.c_uint64), ("max_stack", C.c_uint32), ("pad", C.c_uint32)]

    def as_dict(self):
        n = max(self.rays, 1)
        return {"rays": self.rays, "inner_nodes": self.inner_nodes, "prim_packets": self.prim_packets,
                "hits": self.hits, "max_stack": self.max_stack,
                "inner_per_ray": self.inner_nodes / n, "prims_per_ray": self.prim_packets / n}


_lib = None


def build():
    srcs = sorted(HERE.glob("*.c"))
    if not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs):
        subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                        "-shared", *map(str, srcs), "-lm", "-o", str(LIB_PATH)], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        for name in ("oracle_bvh2_tri1", "oracle_bvh4_tri4", "oracle_bvh8_tri4", "oracle_gpu_bvh4_tri4", "oracle_gpu_bvh8_tri4"):
            fn = getattr(_lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(OracleStats)]
        for name in ("oracle_brute_force_tri1", "oracle_brute_force_tri4"):
            fn = getattr(_lib, name)
            fn.restype = None
            fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        _lib.oracle_abi_sizes.restype = C.c_uint32
        _lib.oracle_abi_sizes.argtypes = [C.c_int]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def traverse(width, nodes, tris, rays, any_hit=False, algo="ref"):
    """Runs the reference algorithm for the given layout.  Returns (hits, stats dict).

    algo="ref": 2 = GPU single-ray BVH2/Tri1 (B1); 4 / 8 = CPU single-ray BVH4/BVH8 + Tri4 (B2).
    algo="gpu": 4 / 8 = the reference GPU kernel's general-arity branch on Node4/Node8 + Tri4 (B1g)."""
    if algo == "gpu" and width != 2:
        fn = {4: "oracle_gpu_bvh4_tri4", 8: "oracle_gpu_bvh8_tri4"}[width]
    else:
        fn = {2: "oracle_bvh2_tri1", 4: "oracle_bvh4_tri4", 8: "oracle_bvh8_tri4"}[width]
    nodes = np.ascontiguousarray(nodes)
    tris = np.ascontiguousarray(tris)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT1)
    st = OracleStats()
    rc = getattr(lib(), fn)(_ptr(nodes), _ptr(tris), _ptr(rays), _ptr(hits), len(rays), int(any_hit), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle traversal stack overflow")
    return hits, st.as_dict()


def ray_steps(nodes, tris, rays, any_hit=False):
    """Per-ray visit counts of B1 (BVH2/Tri1): array (n, 2) = inner nodes, triangles.  Analysis aid."""
    rays = np.ascontiguousarray(rays)
    buf = np.zeros((len(rays), 2), np.uint32)
    l = lib()
    l.oracle_set_ray_step_trace.restype = None
    l.oracle_set_ray_step_trace.argtypes = [C.c_void_p]
    l.oracle_set_ray_step_trace(_ptr(buf))
    try:
        traverse(2, nodes, tris, rays, any_hit=any_hit)
    finally:
        l.oracle_set_ray_step_trace(None)
    return buf


def node_visits(nodes, tris, rays, any_hit=False):
    """Visit count of every inner node under B1 (BVH2/Tri1).  Analysis aid (scripts/model_top_levels.py)."""
    rays = np.ascontiguousarray(rays)
    buf = np.zeros(len(nodes), np.uint64)
    l = lib()
    l.oracle_set_node_visit_trace.restype = None
    l.oracle_set_node_visit_trace.argtypes = [C.c_void_p]
    l.oracle_set_node_visit_trace(_ptr(buf))
    try:
        traverse(2, nodes, tris, rays, any_hit=any_hit)
    finally:
        l.oracle_set_node_visit_trace(None)
    return buf


def ray_depths(nodes, tris, rays, any_hit=False):
    """Deepest stack pointer of every ray under B1 (BVH2/Tri1), uint8.  Analysis aid (scripts/model_stack_depth.py)."""
    rays = np.ascontiguousarray(rays)
    buf = np.zeros(len(rays), np.uint8)
    l = lib()
    l.oracle_set_ray_depth_trace.restype = None
    l.oracle_set_ray_depth_trace.argtypes = [C.c_void_p]
    l.oracle_set_ray_depth_trace(_ptr(buf))
    try:
        traverse(2, nodes, tris, rays, any_hit=any_hit)
    finally:
        l.oracle_set_ray_depth_trace(None)
    return buf


def brute_force(tris, rays):
    """Every ray against every triangle.  Returns (hits, second_t)."""
    tris = np.ascontiguousarray(tris)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT1)
    second = np.zeros(len(rays), "<f4")
    fn = lib().oracle_brute_force_tri1 if tris.dtype.itemsize == 48 else lib().oracle_brute_force_tri4
    fn(_ptr(tris), len(tris), _ptr(rays), _ptr(hits), _ptr(second), len(rays))
    return hits, second


# ---- renderer oracle (oracle/render_oracle.c) ------------------------------------------------

class _Scene(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vertices", "normals", "face_normals", "indices", "nodes", "tris", "materials", "lights",
        "light_ids")] + \
               [("num_tris", C.c_int32), ("num_materials", C.c_int32), ("num_lights", C.c_int32), ("pad", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("texcoords", "textures", "texels")]


class _Settings(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("w", C.c_float),
        ("h", C.c_float)]


def _scene_struct(scene):
    keep = [np.ascontiguousarray(getattr(scene, n)) for n in ("vertices", "normals", "face_normals", "indices", "nodes", "tris",
        "materials", "lights", "light_ids")]
    tex = [np.ascontiguousarray(getattr(scene, n)) for n in ("texcoords", "textures", "texels")]
    s = _Scene(*[_ptr(a) for a in keep], scene.num_tris, len(scene.materials), len(scene.lights), 0, *[_ptr(a) for a in tex])
    return s, keep + tex


def render(scene, cam, iter_, spp, max_path_len, width, height, film=None, rows=None, threads=8):
    """CPU path tracer (one path at a time).  Accumulates into `film` (h, w, 3) float32 and returns
    (film, [primary rays, shadow rays]).  Row bands run on `threads` host threads."""
    import concurrent.futures as cf
    l = lib()
    l.oracle_render.restype = None
    l.oracle_render.argtypes = [C.POINTER(_Scene), C.POINTER(_Settings), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    if film is None:
        film = np.zeros((height, width, 3), "<f4")
    s, keep = _scene_struct(scene)
    st = _Settings((C.c_float * 3)(*cam["eye"]), (C.c_float * 3)(*cam["dir"]), (C.c_float * 3)(*cam["up"]), (C.c_float * 3)(*cam["right"]),
                   float(cam["w"]), float(cam["h"]))
    y0, y1 = rows if rows else (0, height)
    bands = np.linspace(y0, y1, max(1, min(threads, y1 - y0)) + 1).astype(int)
    counts = np.zeros((len(bands) - 1, 2), np.uint64)

    def work(k):
        l.oracle_render(C.byref(s), C.byref(st), iter_, spp, max_path_len, width, height, int(bands[k]), int(bands[k + 1]),
                        _ptr(film), counts[k].ctypes.data_as(C.c_void_p))
    with cf.ThreadPoolExecutor(max(1, threads)) as ex:
        list(ex.map(work, range(len(bands) - 1)))
    return film, counts.sum(axis=0)


def tex_lookup(scene, tex, uv):
    """Bilinear / repeat lookup of texture `tex` at uv (n, 2) -> (n, 3) float32."""
    l = lib()
    s, keep = _scene_struct(scene)
    uv = np.ascontiguousarray(uv, "<f4"); out = np.zeros((len(uv), 3), "<f4")
    l.oracle_tex_lookup.restype = None
    l.oracle_tex_lookup.argtypes = [C.POINTER(_Scene), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    l.oracle_tex_lookup(C.byref(s), int(tex), _ptr(uv), _ptr(out), len(uv))
    return out


def hit_material(scene, prim, u, v):
    """The material table entry a hit on triangle `prim` at barycentrics (u, v) shades with (textures resolved)."""
    l = lib()
    s, keep = _scene_struct(scene)
    out = np.zeros(1, scene.materials.dtype)
    l.oracle_hit_material.restype = None
    l.oracle_hit_material.argtypes = [C.POINTER(_Scene), C.c_int32, C.c_float, C.c_float, C.c_void_p]
    l.oracle_hit_material(C.byref(s), int(prim), float(u), float(v), _ptr(out))
    return out[0]


def tonemap(film, iters):
    """(film / iter)^(1/2.2), clamp, x255 (src/driver/driver.cpp:144-157)."""
    x = np.clip(np.power(np.maximum(film / np.float32(iters), 0), np.float32(1 / 2.2)), 0, 1)
    return (x * 255.0).astype(np.uint8)


# ---- CPU baseline (oracle/hybrid_baseline.cpp): Rodent's hybrid ray8 x bvh8 kernel restated with AVX2 ----
_base = None


def baseline_lib():
    global _base
    if _base is None:
        out = HERE / "libcpu_baseline.so"
        src = HERE / "hybrid_baseline.cpp"
        # liboracle.so first: the baseline library links it (shading of the CPU wavefront renderer)
        lib()
        deps = [src, HERE / "cpu_wavefront.inc", LIB_PATH]
        if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
            subprocess.run(["g++", "-O3", "-std=c++17", "-march=x86-64-v3", "-fPIC", "-shared", "-pthread", str(src), f"-L{HERE}",
                "-l:liboracle.so", "-Wl,-rpath,$ORIGIN",
                            "-o", str(out)], check=True)
        _base = C.CDLL(str(out))
        _base.cpu_baseline_traverse.restype = None
        _base.cpu_baseline_traverse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        _base.cpu_baseline_hardware_threads.restype = C.c_int32
    return _base


def cpu_baseline(nodes8, tris4, rays, any_hit=False, mode="hybrid", threads=1):
    """mode "hybrid" (ray8 x bvh8 packets with single-ray fallback) or "single".  Rays beyond a multiple of 8 are dropped."""
    nodes8 = np.ascontiguousarray(nodes8); tris4 = np.ascontiguousarray(tris4); rays = np.ascontiguousarray(rays)
    n = len(rays) // 8 * 8
    hits = np.zeros(n, HIT1)
    baseline_lib().cpu_baseline_traverse(_ptr(nodes8), _ptr(tris4), _ptr(rays), _ptr(hits), n, int(any_hit), 0 if mode == "hybrid" else 1,
        int(threads))
    return hits


def cpu_baseline_bench(nodes8, tris4, rays, threads, passes, any_hit=False, mode="hybrid"):
    """`passes` timed passes (after one warm-up pass) on a persistent pool of `threads` threads; returns (seconds per pass, hits)."""
    nodes8 = np.ascontiguousarray(nodes8); tris4 = np.ascontiguousarray(tris4); rays = np.ascontiguousarray(rays)
    n = len(rays) // 8 * 8
    hits = np.zeros(n, HIT1)
    secs = np.zeros(passes, np.float64)
    l = baseline_lib()
    l.cpu_baseline_bench.restype = None
    l.cpu_baseline_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
        C.c_void_p]
    l.cpu_baseline_bench(_ptr(nodes8), _ptr(tris4), _ptr(rays), _ptr(hits), n, int(any_hit), 0 if mode == "hybrid" else 1, int(threads),
        int(passes), _ptr(secs))
    return secs, hits


def render_wavefront(scene, nodes8, tris4, cam, iter_, spp, max_path_len, width, height, film=None, threads=8):
    """CPU BASELINE for frames: the reference's tile-parallel wavefront mapping (mapping_cpu.impala:352-473) over the hybrid
    ray8 x BVH8 traversal (oracle/cpu_wavefront.inc).  Returns (film, [primary rays, shadow rays])."""
    l = baseline_lib()
    l.cpu_wavefront_render.restype = None
    l.cpu_wavefront_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(_Settings), C.c_int32,
        C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_void_p]
    if film is None:
        film = np.zeros((height, width, 3), "<f4")
    s, keep = _scene_struct(scene)
    nodes8 = np.ascontiguousarray(nodes8); tris4 = np.ascontiguousarray(tris4)
    indices = np.ascontiguousarray(scene.indices)
    st = _Settings((C.c_float * 3)(*cam["eye"]), (C.c_float * 3)(*cam["dir"]), (C.c_float * 3)(*cam["up"]), (C.c_float * 3)(*cam["right"]),
        float(cam["w"]), float(cam["h"]))
    counts = np.zeros(2, np.uint64)
    l.cpu_wavefront_render(C.byref(s), _ptr(indices), len(scene.materials), _ptr(nodes8), _ptr(tris4), C.byref(st), iter_, spp,
        max_path_len, width, height,
                           _ptr(film), int(threads), _ptr(counts))
    return film, counts


def hardware_threads():
    return int(baseline_lib().cpu_baseline_hardware_threads())
