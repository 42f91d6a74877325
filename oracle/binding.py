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


def brute_force(tris, rays):
    """Every ray against every triangle.  Returns (hits, second_t)."""
    tris = np.ascontiguousarray(tris)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT1)
    second = np.zeros(len(rays), "<f4")
    fn = lib().oracle_brute_force_tri1 if tris.dtype.itemsize == 48 else lib().oracle_brute_force_tri4
    fn(_ptr(tris), len(tris), _ptr(rays), _ptr(hits), _ptr(second), len(rays))
    return hits, second
