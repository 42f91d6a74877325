"""The scene-data services of the renderer ABI (include/rodent_render.h; reference: src/driver/interface.cpp:432-492,
584-629,665-673): the reference converter's data files go through rodent_load_* onto the device and are traced /
read back from there; host stream slabs keep the reference's carving."""
import ctypes as C
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from rodent_amd import formats as F, scene as S


class _Ptr:
    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


class _LoadedBvh:
    """Duck type of abi.DeviceBvh around pointers the library owns."""
    def __init__(self, width, nodes, tris, dev=0):
        self.width, self.dev, self.nodes, self.tris = width, dev, _Ptr(nodes), _Ptr(tris)


def test_cpu_stream_slabs_keep_the_reference_carving(native_build):
    """rodent_cpu_get_{primary,secondary}_stream (interface.cpp:341-342,367-373,621-629): one slab, array k at ptr + k * capacity,
    capacity rounded to (size & ~31) + 32; no GPU involved."""
    from rodent_amd import render as R
    l = R.lib()
    l.rodent_cpu_get_primary_stream.argtypes = [C.POINTER(R.PrimaryStream), C.c_int32]; l.rodent_cpu_get_primary_stream.restype = None
    l.rodent_cpu_get_secondary_stream.argtypes = [C.POINTER(R.SecondaryStream), C.c_int32]; l.rodent_cpu_get_secondary_stream.restype = None
    p, s = R.PrimaryStream(), R.SecondaryStream()
    l.rodent_cpu_get_primary_stream(C.byref(p), 1000)
    l.rodent_cpu_get_secondary_stream(C.byref(s), 1000)
    cap = (1000 & ~31) + 32
    base = p.rays.id
    names = ["org_x", "org_y", "org_z", "dir_x", "dir_y", "dir_z", "tmin", "tmax"]
    assert [getattr(p.rays, n) - base for n in names] == [4 * cap * (k + 1) for k in range(8)]
    assert [getattr(p, n) - base for n in ("geom_id", "prim_id", "t", "u", "v", "rnd", "mis", "contrib_r", "contrib_g", "contrib_b",
        "depth")] == [4 * cap * k for k in range(9, 20)]
    assert [getattr(s, n) - s.rays.id for n in ("prim_id", "color_r", "color_g", "color_b")] == [4 * cap * k for k in range(9, 13)]
    # host memory: writable from here
    C.memset(base, 0, 4 * cap * 20)
    p2 = R.PrimaryStream(); l.rodent_cpu_get_primary_stream(C.byref(p2), 500)      # smaller request: same slab (interface.cpp:359-366)
    assert p2.rays.id == base
    t0 = l.clock_us(); t1 = l.clock_us()
    assert 0 <= t1 - t0 < 1_000_000


@pytest.fixture(scope="module")
def data_dir(native_build, tmp_path_factory):
    d = tmp_path_factory.mktemp("svc")
    data = d / "data"; data.mkdir()
    subprocess.run([native_build.BIN_DIR / "converter", GOLDEN / "cornell_box.obj", "-o", d / "c.rscene", "--data-dir", data], check=True,
        capture_output=True)
    # the reference's CPU targets also store BVH4 / BVH8 layouts in bvh.bin (converter.cpp:428-438): append them
    for block in (F.BVH4_TRI4, F.BVH8_TRI4):
        n, t = F.read_bvh(GOLDEN / "cornell.bvh", block)
        F.write_bvh_bin(data / "bvh.bin", n, t, append=True)
    return d, S.Scene(d / "c.rscene")


@pytest.mark.gpu
def test_loaded_bvhs_are_traced_from_device_memory(native_build, oracle, data_dir, cornell):
    import torch
    from rodent_amd import abi, render as R
    assert torch.cuda.is_available()
    d, scene = data_dir
    l = R.lib()
    path = str(d / "data" / "bvh.bin").encode()
    rays = cornell.ray_sets["primary"]
    host = {2: (scene.nodes, scene.tris), 4: cornell.blocks[4], 8: cornell.blocks[8]}
    algo = {2: "ref", 4: "gpu", 8: "gpu"}
    for width, fn in ((2, l.rodent_load_bvh2_tri1), (4, l.rodent_load_bvh4_tri4), (8, l.rodent_load_bvh8_tri4)):
        nodes, tris = C.c_void_p(), C.c_void_p()
        fn(0, path, C.byref(nodes), C.byref(tris))
        again_n, again_t = C.c_void_p(), C.c_void_p()
        fn(0, path, C.byref(again_n), C.byref(again_t))
        assert (again_n.value, again_t.value) == (nodes.value, tris.value)          # cached by (dev, file) (interface.cpp:395-423)
        nn, nt = C.c_int32(), C.c_int32()
        l.rodent_hip_bvh_counts(0, path, width, C.byref(nn), C.byref(nt))
        assert (nn.value, nt.value) == (len(host[width][0]), len(host[width][1]))
        bvh = _LoadedBvh(width, nodes.value, tris.value)
        rd = abi.to_device(rays, 0)
        hd = torch.zeros(len(rays) * 16, dtype=torch.uint8, device="cuda:0")
        abi.traverse_async(bvh, rd, hd, len(rays), False, 0)
        torch.cuda.synchronize()
        ref, _ = oracle.traverse(width, *host[width], rays, algo=algo[width])
        assert abi.from_device(hd, F.HIT1).tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_loaded_buffers_and_images(native_build, data_dir, textured_scene):
    import torch
    from rodent_amd import render as R
    d, scene = data_dir
    l = R.lib()
    for name, arr in (("vertices", scene.vertices), ("indices", scene.indices), ("light_ids", scene.light_ids)):
        path = str(d / "data" / f"{name}.bin").encode()
        p = l.rodent_load_buffer(0, path)
        assert p and l.rodent_load_buffer(0, path) == p
        size = l.rodent_hip_buffer_size(0, path)
        assert size == arr.nbytes
        back = R.read_stream_array(p, size // 4, arr.dtype.base if arr.dtype.fields is None else np.uint32)
        assert back.tobytes() == np.ascontiguousarray(arr).tobytes(), name
    assert l.rodent_hip_buffer_size(0, b"/nonexistent.bin") == -1
    # textures: the texel pool of the converted textured scene holds the same decoded images (image.cpp: RGBA8, flipped, gamma)
    tscene, tdir = textured_scene
    for k, (fname, fn) in enumerate((("checker.png", l.rodent_load_png), ("grad.jpg", l.rodent_load_jpg))):
        px, w, h = C.c_void_p(), C.c_int32(), C.c_int32()
        fn(0, str(tdir / fname).encode(), C.byref(px), C.byref(w), C.byref(h))
        match = [t for t in tscene.textures if (t["width"], t["height"]) == (w.value, h.value)]
        assert match, fname
        texels = R.read_stream_array(px.value, w.value * h.value, np.uint32)
        assert any(np.array_equal(texels, tscene.texels[t["offset"]: t["offset"] + w.value * h.value]) for t in match), fname
    R.lib().cleanup_interface()                                   # frees what the services loaded (interface.cpp:516-518)
    assert l.rodent_hip_buffer_size(0, str(d / "data" / "vertices.bin").encode()) == -1


def test_missing_files_abort_like_the_reference(native_build, tmp_path):
    """error() in interface.cpp:436,462 prints and aborts: run in a child process."""
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from rodent_amd import render as R; l = R.lib(); "
            "n, t = ctypes.c_void_p(), ctypes.c_void_p(); l.rodent_load_bvh2_tri1(0, b'/nonexistent/bvh.bin', ctypes.byref(n), "
                "ctypes.byref(t))") % str(GOLDEN.parents[1])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot open BVH" in r.stderr
