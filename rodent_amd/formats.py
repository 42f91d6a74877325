"""numpy views of Rodent's on-disk formats and traversal structs.

Struct layouts are the ones in include/rodent_traversal.h (bit-compatible with
the reference: src/traversal/mapping_gpu.impala:3-16, mapping_cpu.impala:3-22,
tools/bench_traversal/bench_traversal.impala:25-65).  File formats follow
tools/common/load_bvh.h:21-74 (.bvh), tools/common/load_rays.h:59-92 (.rays)
and tools/bench_traversal/bench_traversal.cpp:342-346 (.fbuf).
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

BVH_MAGIC = 0x95CBED1F
BVH2_TRI1, BVH4_TRI4, BVH8_TRI4 = 1, 2, 3

NODE2 = np.dtype([("bounds", "<f4", (12,)), ("child", "<i4", (2,)), ("pad", "<i4", (2,))])
TRI1 = np.dtype([("v0", "<f4", (3,)), ("pad", "<i4"), ("e1", "<f4", (3,)), ("geom_id", "<i4"),
                 ("e2", "<f4", (3,)), ("prim_id", "<i4")])
NODE4 = np.dtype([("bounds", "<f4", (6, 4)), ("child", "<i4", (4,)), ("pad", "<i4", (4,))])
NODE8 = np.dtype([("bounds", "<f4", (6, 8)), ("child", "<i4", (8,)), ("pad", "<i4", (8,))])
TRI4 = np.dtype([("v0", "<f4", (3, 4)), ("e1", "<f4", (3, 4)), ("e2", "<f4", (3, 4)), ("n", "<f4", (3, 4)),
                 ("prim_id", "<i4", (4,)), ("geom_id", "<i4", (4,))])
RAY1 = np.dtype([("org", "<f4", (3,)), ("tmin", "<f4"), ("dir", "<f4", (3,)), ("tmax", "<f4")])
HIT1 = np.dtype([("tri_id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

assert (NODE2.itemsize, TRI1.itemsize, NODE4.itemsize, NODE8.itemsize, TRI4.itemsize,
        RAY1.itemsize, HIT1.itemsize) == (64, 48, 128, 256, 224, 32, 16)

_BLOCK_DTYPES = {BVH2_TRI1: (NODE2, TRI1), BVH4_TRI4: (NODE4, TRI4), BVH8_TRI4: (NODE8, TRI4)}


def read_bvh(path, block_type):
    """Returns (nodes, tris) structured arrays of the requested block.

    Raises ValueError on a bad magic / missing block (the reference's load_bvh
    returns false and the tool prints "Cannot load BVH file")."""
    node_dt, tri_dt = _BLOCK_DTYPES[block_type]
    data = Path(path).read_bytes()
    if len(data) < 4 or struct.unpack_from("<I", data, 0)[0] != BVH_MAGIC:
        raise ValueError(f"{path}: not a .bvh file (bad magic)")
    pos = 4
    while pos + 12 <= len(data):
        offset, btype = struct.unpack_from("<QI", data, pos)
        if btype == block_type:
            n_nodes, n_tris = struct.unpack_from("<II", data, pos + 12)
            start = pos + 20
            need = node_dt.itemsize * n_nodes + tri_dt.itemsize * n_tris
            if offset != 12 + need or start + need > len(data):
                raise ValueError(f"{path}: truncated or inconsistent block {block_type}")
            nodes = np.frombuffer(data, node_dt, n_nodes, start).copy()
            tris = np.frombuffer(data, tri_dt, n_tris, start + node_dt.itemsize * n_nodes).copy()
            return nodes, tris
        pos += 8 + offset
    raise ValueError(f"{path}: no block of type {block_type}")


def write_bvh(path, blocks):
    """blocks: iterable of (block_type, nodes, tris)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", BVH_MAGIC))
        for btype, nodes, tris in blocks:
            node_dt, tri_dt = _BLOCK_DTYPES[btype]
            nodes = np.ascontiguousarray(nodes, node_dt)
            tris = np.ascontiguousarray(tris, tri_dt)
            f.write(struct.pack("<QIII", 12 + nodes.nbytes + tris.nbytes, btype, len(nodes), len(tris)))
            f.write(nodes.tobytes())
            f.write(tris.tobytes())


def read_rays(path, tmin=0.0, tmax=1e9):
    """.rays -> Ray1 array with the given [tmin, tmax] (defaults: bench_traversal.cpp:141)."""
    if Path(path).stat().st_size % 24:
        raise ValueError(f"{path}: size is not a multiple of 24 bytes")
    raw = np.fromfile(path, "<f4")
    raw = raw.reshape(-1, 6)
    return make_rays(raw[:, :3], raw[:, 3:], tmin, tmax)


def make_rays(org, direction, tmin=0.0, tmax=1e9):
    org = np.asarray(org, "<f4").reshape(-1, 3)
    rays = np.empty(len(org), RAY1)
    rays["org"] = org
    rays["dir"] = np.asarray(direction, "<f4").reshape(-1, 3)
    rays["tmin"] = tmin
    rays["tmax"] = tmax
    return rays


def write_rays(path, rays):
    out = np.empty((len(rays), 6), "<f4")
    out[:, :3] = rays["org"]
    out[:, 3:] = rays["dir"]
    out.tofile(path)


def write_fbuf(path, hits):
    np.ascontiguousarray(hits["t"], "<f4").tofile(path)


def read_fbuf(path):
    return np.fromfile(path, "<f4")


# ---- the reference's LZ4 buffer files (src/driver/buffer.h): [u32 size][u32 compressed size][LZ4 block] -------------
# Python side of the data/*.bin compatibility: liblz4 through ctypes (the C++ side carries its own block codec,
# rodent_amd/host/lz4_block.h, because lz4.h is not in this image); the tests play the two against each other.
_lz4 = None


def _liblz4():
    global _lz4
    if _lz4 is None:
        import ctypes as C
        _lz4 = C.CDLL("liblz4.so.1")
        _lz4.LZ4_compressBound.restype = C.c_int; _lz4.LZ4_compressBound.argtypes = [C.c_int]
        _lz4.LZ4_compress_default.restype = C.c_int; _lz4.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        _lz4.LZ4_decompress_safe.restype = C.c_int; _lz4.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    return _lz4


def _read_buffer(data, pos):
    import ctypes as C
    size, csize = struct.unpack_from("<II", data, pos)
    out = C.create_string_buffer(max(size, 1))
    n = _liblz4().LZ4_decompress_safe(data[pos + 8: pos + 8 + csize], out, csize, size)
    if n != size:
        raise ValueError("corrupt LZ4 buffer")
    return out.raw[:size], pos + 8 + csize


def _pack_buffer(raw):
    import ctypes as C
    bound = _liblz4().LZ4_compressBound(len(raw))
    out = C.create_string_buffer(max(bound, 1))
    n = _liblz4().LZ4_compress_default(raw, out, len(raw), bound)
    return struct.pack("<II", len(raw), n) + out.raw[:n]


def read_buffer_file(path, dtype=np.uint8):
    raw, _ = _read_buffer(Path(path).read_bytes(), 0)
    return np.frombuffer(raw, dtype).copy()


def write_buffer_file(path, array):
    Path(path).write_bytes(_pack_buffer(np.ascontiguousarray(array).tobytes()))


def read_bvh_bin(path, node_dtype=None, tri_dtype=None):
    """data/bvh.bin (converter.cpp:428-437): the layout whose element sizes match (default: BVH2/Tri1) -> (nodes, tris)."""
    node_dtype = node_dtype or NODE2; tri_dtype = tri_dtype or TRI1
    data = Path(path).read_bytes()
    pos = 0
    while pos + 8 <= len(data):
        ns, ts = struct.unpack_from("<II", data, pos)
        pos += 8
        nodes, pos = _read_buffer(data, pos)
        tris, pos = _read_buffer(data, pos)
        if (ns, ts) == (node_dtype.itemsize, tri_dtype.itemsize):
            return np.frombuffer(nodes, node_dtype).copy(), np.frombuffer(tris, tri_dtype).copy()
    raise ValueError(f"{path}: no BVH with {node_dtype.itemsize}-byte nodes and {tri_dtype.itemsize}-byte triangles")


def write_bvh_bin(path, nodes, tris, append=False):
    with open(path, "ab" if append else "wb") as f:
        f.write(struct.pack("<II", nodes.dtype.itemsize,
            tris.dtype.itemsize) + _pack_buffer(nodes.tobytes()) + _pack_buffer(tris.tobytes()))
