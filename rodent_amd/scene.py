"""Reader of the `.rscene` scene-description file written by rodent_amd/bin/converter
(host/scene.cpp): mesh, BVH2, material and light tables as numpy arrays."""
from __future__ import annotations

import struct
import subprocess
from pathlib import Path

import numpy as np

from . import build, formats as F

MATERIAL = np.dtype([("kd", "<f4", (3,)), ("type", "<i4"), ("ks", "<f4", (3,)), ("ns", "<f4"), ("tf", "<f4", (3,)), ("ni", "<f4"),
                     ("mix_k", "<f4"), ("emissive", "<i4"), ("tex_kd", "<i4"), ("tex_ks", "<i4")])
TEXTURE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<u4"), ("pad", "<i4")])
LIGHT = np.dtype([("v0", "<f4", (4,)), ("v1", "<f4", (4,)), ("v2", "<f4", (4,)), ("n", "<f4", (3,)), ("inv_area", "<f4"),
    ("color", "<f4", (4,))])
assert MATERIAL.itemsize == 64 and LIGHT.itemsize == 80
MAGIC = 0x43534452


class Scene:
    """Arrays of one scene; field names follow RodentSceneDesc (include/rodent_render.h)."""

    def __init__(self, path):
        data = Path(path).read_bytes()
        hdr = struct.unpack_from("<12I", data, 0)
        if hdr[0] != MAGIC or hdr[1] != 2:
            raise ValueError(f"{path}: not a .rscene file (version 2)")
        self.default_spp, self.default_max_path_len = hdr[2], hdr[3]
        nv, nt, nn, nbt, nm, nl, ntex, ntexels = hdr[4:12]
        pos = 48

        def take(dtype, count):
            nonlocal pos
            a = np.frombuffer(data, dtype, count, pos).copy()
            pos += a.nbytes
            return a
        self.vertices = take("<f4", 4 * nv).reshape(-1, 4)
        self.normals = take("<f4", 4 * nv).reshape(-1, 4)
        self.face_normals = take("<f4", 4 * nt).reshape(-1, 4)
        self.indices = take("<i4", 4 * nt).reshape(-1, 4)
        self.nodes = take(F.NODE2, nn)
        self.tris = take(F.TRI1, nbt)
        self.materials = take(MATERIAL, nm)
        self.lights = take(LIGHT, nl)
        self.light_ids = take("<i4", nt)
        self.texcoords = take("<f4", 4 * nv).reshape(-1, 4)
        self.textures = take(TEXTURE, ntex)
        self.texels = take("<u4", ntexels)
        if pos != len(data):
            raise ValueError(f"{path}: trailing bytes")

    @property
    def num_tris(self):
        return len(self.indices)


def convert(obj_path, out_path, spp=4, max_path_len=64):
    """Runs the converter tool (built on demand)."""
    tool = build.BIN_DIR / "converter"
    if not tool.exists():
        build.build_host()
    subprocess.run([str(tool), str(obj_path), "-o", str(out_path), "-spp", str(spp), "--max-path-len", str(max_path_len)],
                   check=True, stdout=subprocess.DEVNULL)
    return Scene(out_path)


def camera_settings(eye, direction, up, fov, width, height):
    """Settings record as src/driver/driver.cpp:29-39,286-296 fills it."""
    f32 = np.float32
    d = np.asarray(direction, f32); d = d / np.sqrt((d * d).sum(dtype=f32), dtype=f32)
    r = np.cross(d, np.asarray(up, f32)).astype(f32); r = r / np.sqrt((r * r).sum(dtype=f32), dtype=f32)
    u = np.cross(r, d).astype(f32); u = u / np.sqrt((u * u).sum(dtype=f32), dtype=f32)
    w = f32(np.tan(f32(fov) * f32(3.14159265359) / f32(360.0)))
    h = f32(w / f32(width / height))
    return {"eye": np.asarray(eye, f32), "dir": d, "up": u, "right": r, "w": w, "h": h}
