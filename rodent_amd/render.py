"""ctypes binding of the renderer ABI (include/rodent_render.h): scene upload, render(), film access.

Mirrors what src/driver/driver.cpp does around the generated `render()`:
setup_interface -> (clear_pixels) -> render(settings, iter++) ... -> get_pixels."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi

i32, vp = C.c_int32, C.c_void_p


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Settings(C.Structure):
    _fields_ = [("eye", Vec3), ("dir", Vec3), ("up", Vec3), ("right", Vec3), ("width", C.c_float), ("height", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [(n, vp) for n in ("vertices", "normals", "face_normals", "indices", "nodes", "tris", "materials", "lights",
        "light_ids")] + \
               [(n, i32) for n in ("num_vertices", "num_tris", "num_nodes", "num_bvh_tris", "num_materials", "num_lights")] + \
               [(n, vp) for n in ("texcoords", "textures", "texels")] + [("num_textures", i32), ("num_texels", C.c_uint32)]


RENDER_EXPORTS = ["rodent_hip_scene_create", "rodent_hip_scene_destroy", "rodent_hip_render_config", "rodent_hip_render_mapping",
    "rodent_hip_render_capacity", "rodent_hip_render_sort", "rodent_hip_render_hit_records", "rodent_hip_render_overlap",
    "rodent_hip_render_fused_sort", "rodent_hip_render_fused_compact", "rodent_hip_render_mapping_in_effect", "rodent_hip_render_defaults",
    "rodent_hip_render_lds_image", "rodent_hip_render_mega_joint", "rodent_hip_render_trace_persistent", "rodent_hip_render_trace_refill",
    "rodent_hip_render_trace_refill_in_effect", "get_spp", "render",
                  "setup_interface", "get_pixels", "clear_pixels", "cleanup_interface", "rodent_get_film_data",
                  "rodent_gpu_get_first_primary_stream", "rodent_gpu_get_second_primary_stream", "rodent_gpu_get_secondary_stream",
                  "rodent_gpu_get_tmp_buffer", "rodent_present", "rodent_hip_set_device", "rodent_hip_render_rows",
                      "rodent_hip_render_tiles",
                  "rodent_hip_render_counters", "hip_generate_rays", "hip_traverse_primary", "hip_sort_primary", "hip_shade",
                  "hip_traverse_secondary", "hip_compact_primary",
                  "rodent_load_buffer", "rodent_load_bvh2_tri1", "rodent_load_bvh4_tri4", "rodent_load_bvh8_tri4", "rodent_load_png",
                      "rodent_load_jpg",
                  "rodent_cpu_get_primary_stream", "rodent_cpu_get_secondary_stream", "clock_us", "rodent_hip_buffer_size",
                      "rodent_hip_bvh_counts"]

_ready = False


def lib():
    global _ready
    l = abi.lib()
    if not _ready:
        l.rodent_hip_scene_create.argtypes = [i32, C.POINTER(SceneDesc)]; l.rodent_hip_scene_create.restype = None
        l.rodent_hip_scene_destroy.argtypes = [i32]; l.rodent_hip_scene_destroy.restype = None
        l.rodent_hip_render_config.argtypes = [i32, i32, i32]; l.rodent_hip_render_config.restype = None
        l.rodent_hip_render_mapping.argtypes = [i32, i32]; l.rodent_hip_render_mapping.restype = None
        l.rodent_hip_render_capacity.argtypes = [i32, i32]; l.rodent_hip_render_capacity.restype = None
        l.rodent_hip_render_sort.argtypes = [i32, i32]; l.rodent_hip_render_sort.restype = None
        l.rodent_hip_render_overlap.argtypes = [i32, i32]; l.rodent_hip_render_overlap.restype = None
        l.rodent_hip_render_hit_records.argtypes = [i32, i32]; l.rodent_hip_render_hit_records.restype = None
        l.rodent_hip_render_fused_sort.argtypes = [i32, i32]; l.rodent_hip_render_fused_sort.restype = None
        l.rodent_hip_render_lds_image.argtypes = [i32, i32]; l.rodent_hip_render_lds_image.restype = None
        l.rodent_hip_render_fused_compact.argtypes = [i32, i32]; l.rodent_hip_render_fused_compact.restype = None
        l.rodent_hip_render_mapping_in_effect.argtypes = [i32]; l.rodent_hip_render_mapping_in_effect.restype = i32
        l.rodent_hip_render_defaults.argtypes = [i32]; l.rodent_hip_render_defaults.restype = None
        l.rodent_hip_render_trace_persistent.argtypes = [i32, i32]; l.rodent_hip_render_trace_persistent.restype = None
        l.rodent_hip_render_trace_refill.argtypes = [i32, i32, i32]; l.rodent_hip_render_trace_refill.restype = None
        l.rodent_hip_render_trace_refill_in_effect.argtypes = [i32]; l.rodent_hip_render_trace_refill_in_effect.restype = i32
        l.rodent_hip_render_mega_joint.argtypes = [i32, i32]; l.rodent_hip_render_mega_joint.restype = None
        l.get_spp.argtypes = []; l.get_spp.restype = i32
        l.render.argtypes = [C.POINTER(Settings), i32]; l.render.restype = None
        l.setup_interface.argtypes = [C.c_size_t, C.c_size_t]; l.setup_interface.restype = None
        l.get_pixels.argtypes = []; l.get_pixels.restype = C.POINTER(C.c_float)
        l.clear_pixels.argtypes = []; l.clear_pixels.restype = None
        l.cleanup_interface.argtypes = []; l.cleanup_interface.restype = None
        l.rodent_present.argtypes = [i32]; l.rodent_present.restype = None
        l.rodent_hip_set_device.argtypes = [i32]; l.rodent_hip_set_device.restype = None
        l.rodent_hip_render_rows.argtypes = [i32, C.POINTER(Settings), i32, i32, i32, vp]; l.rodent_hip_render_rows.restype = None
        l.rodent_hip_render_tiles.argtypes = [i32, C.POINTER(Settings), i32, i32, i32, i32, vp]; l.rodent_hip_render_tiles.restype = None
        l.rodent_hip_render_counters.argtypes = [i32, C.POINTER(C.c_uint64)]; l.rodent_hip_render_counters.restype = None
        l.rodent_load_buffer.argtypes = [i32, C.c_char_p]; l.rodent_load_buffer.restype = vp
        for name in ("rodent_load_bvh2_tri1", "rodent_load_bvh4_tri4", "rodent_load_bvh8_tri4"):
            fn = getattr(l, name); fn.argtypes = [i32, C.c_char_p, C.POINTER(vp), C.POINTER(vp)]; fn.restype = None
        for name in ("rodent_load_png", "rodent_load_jpg"):
            fn = getattr(l, name); fn.argtypes = [i32, C.c_char_p, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]; fn.restype = None
        l.rodent_hip_buffer_size.argtypes = [i32, C.c_char_p]; l.rodent_hip_buffer_size.restype = C.c_int64
        l.rodent_hip_bvh_counts.argtypes = [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32)]; l.rodent_hip_bvh_counts.restype = None
        l.clock_us.argtypes = []; l.clock_us.restype = C.c_int64
        _ready = True
    return l


def make_settings(cam) -> Settings:
    v = lambda a: Vec3(float(a[0]), float(a[1]), float(a[2]))
    return Settings(v(cam["eye"]), v(cam["dir"]), v(cam["up"]), v(cam["right"]), float(cam["w"]), float(cam["h"]))


class Renderer:
    """One scene on one GPU.  render(cam, iter) accumulates `spp` samples per pixel into the film."""

    MAPPINGS = {"auto": -1, "streaming": 0, "megakernel": 1}       # per scene / mapping_gpu.impala:308-369 / :371-474

    def __init__(self, scene, width, height, spp=4, max_path_len=64, dev=0, mapping="streaming", capacity=0, sort=None, overlap=None,
        fused_sort=None, lds_image=None,
                 trace_persistent=None, fused_compact=None, mega_joint=None, trace_refill=None, hit_records_aos=None):
        """Options left at None take the library's default, or what the option's RODENT_HIP_* environment variable says."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("rodent_amd: no GPU visible (the renderer has no CPU fallback)")
        self.dev, self.width, self.height, self.spp = dev, width, height, spp
        l = lib()
        keep = [np.ascontiguousarray(getattr(scene, n)) for n in ("vertices", "normals", "face_normals", "indices", "nodes", "tris",
            "materials", "lights", "light_ids")]
        tex = [np.ascontiguousarray(getattr(scene, n)) for n in ("texcoords", "textures", "texels")]
        desc = SceneDesc(*[a.ctypes.data_as(vp) for a in keep], len(scene.vertices), scene.num_tris, len(scene.nodes), len(scene.tris),
                         len(scene.materials), len(scene.lights), *[a.ctypes.data_as(vp) for a in tex], len(scene.textures),
                             len(scene.texels))
        l.rodent_hip_set_device(dev)
        l.rodent_hip_render_defaults(dev)                    # options of an earlier Renderer in this process do not leak into this one
        l.rodent_hip_scene_create(dev, C.byref(desc))
        l.rodent_hip_render_config(dev, spp, max_path_len)
        l.rodent_hip_render_mapping(dev, self.MAPPINGS[mapping])
        l.rodent_hip_render_capacity(dev, capacity)          # rays per stream, 0 = default (32 Mi)
        # sort hit rays by material before shading (reference behaviour)
        for value, setter in ((sort, l.rodent_hip_render_sort),
                              (overlap, l.rodent_hip_render_overlap),              # shadow rays on a second HIP stream
                              # hit records of the loop's streams as 20-byte records (default) / the ABI's five arrays
                              (hit_records_aos, l.rodent_hip_render_hit_records),
                              # the sort computes a permutation, the shader gathers through it
                              (fused_sort, l.rodent_hip_render_fused_sort),
                              # stream traversal kernels stage the top of the BVH in LDS
                              (lds_image, l.rodent_hip_render_lds_image),
                              # 0 / 1 / 2: 2-wave kernels + second stream / persistent / joint persistent launch (default: per scene)
                              (trace_persistent, l.rodent_hip_render_trace_persistent),
                              # megakernel: shadow ray + next path ray of a lane in one loop (measured slower) / the reference's sequence of
                              # loops (default)
                              (mega_joint, l.rodent_hip_render_mega_joint),
                              # the shader writes continuing rays to their compacted slots
                              (fused_compact, l.rodent_hip_render_fused_compact)):
            if value is not None:
                setter(dev, int(value))
        # persistent traversal launches: lane refill once that many lanes of a wave are idle: n or (bounce rays, shadow rays); 0 = whole
        # chunks
        if trace_refill is not None:
            bounce, shadow = trace_refill if isinstance(trace_refill, (tuple, list)) else (trace_refill, trace_refill)
            l.rodent_hip_render_trace_refill(dev, int(bounce), int(shadow))
        l.setup_interface(width, height)
        l.clear_pixels()

    def configure(self, spp, max_path_len):
        self.spp = spp
        lib().rodent_hip_render_config(self.dev, spp, max_path_len)

    def mapping_name(self):
        """The mapping the next frame uses ("streaming" / "megakernel"): the caller's choice, or the library's for this scene."""
        return {0: "streaming", 1: "megakernel"}[lib().rodent_hip_render_mapping_in_effect(self.dev)]

    def trace_refill(self):
        """(idle lanes that trigger a refill while a wave draws bounce rays, ... shadow rays) of the persistent traversal launches; (0, 0) =
        whole chunks."""
        v = lib().rodent_hip_render_trace_refill_in_effect(self.dev)
        return v & 255, v >> 8

    def clear(self):
        lib().clear_pixels()

    def render(self, cam, iter_):
        st = make_settings(cam)
        lib().render(C.byref(st), iter_)

    def render_rows(self, cam, iter_, y0, y1):
        st = make_settings(cam)
        lib().rodent_hip_render_rows(self.dev, C.byref(st), iter_, y0, y1, None)

    def render_tiles(self, cam, iter_, tile_rows, first_tile, tile_stride):
        """The interleaved row tiles first_tile, first_tile + tile_stride, ... of tile_rows rows each (GPU k of K: first_tile = k,
        tile_stride = K)."""
        st = make_settings(cam)
        lib().rodent_hip_render_tiles(self.dev, C.byref(st), iter_, tile_rows, first_tile, tile_stride, None)

    def film(self):
        lib().rodent_present(self.dev)
        p = lib().get_pixels()
        return np.ctypeslib.as_array(p, shape=(self.height, self.width, 3)).copy()

    def counters(self):
        buf = (C.c_uint64 * 4)()
        lib().rodent_hip_render_counters(self.dev, buf)
        return {"primary_rays": buf[0], "shadow_rays": buf[1], "iterations": buf[2], "generated": buf[3]}

    def close(self):
        lib().rodent_hip_scene_destroy(self.dev)
        lib().cleanup_interface()


def tonemap(film, iters):
    """(film / iter)^(1/2.2), clamp, x255 -- src/driver/driver.cpp:144-157."""
    x = np.clip(np.power(np.maximum(film / np.float32(iters), 0), np.float32(1 / 2.2)), 0, 1)
    return (x * 255.0).astype(np.uint8)


# ---- stage-level API (the reference's device kernels as separate entry points) ----------------
class RayStream(C.Structure):
    _fields_ = [(n, vp) for n in ("id", "org_x", "org_y", "org_z", "dir_x", "dir_y", "dir_z", "tmin", "tmax")]


class PrimaryStream(C.Structure):
    _fields_ = [("rays", RayStream)] + [(n, vp) for n in ("geom_id", "prim_id", "t", "u", "v", "rnd", "mis", "contrib_r", "contrib_g",
        "contrib_b", "depth")] + \
               [("size", i32), ("pad", i32)]


class SecondaryStream(C.Structure):
    _fields_ = [("rays", RayStream)] + [(n, vp) for n in ("prim_id", "color_r", "color_g", "color_b")] + [("size", i32), ("pad", i32)]


def stage_lib():
    l = lib()
    l.rodent_gpu_get_first_primary_stream.argtypes = [i32, C.POINTER(PrimaryStream),
        i32]; l.rodent_gpu_get_first_primary_stream.restype = None
    l.rodent_gpu_get_second_primary_stream.argtypes = [i32, C.POINTER(PrimaryStream),
        i32]; l.rodent_gpu_get_second_primary_stream.restype = None
    l.rodent_gpu_get_secondary_stream.argtypes = [i32, C.POINTER(SecondaryStream), i32]; l.rodent_gpu_get_secondary_stream.restype = None
    l.rodent_get_film_data.argtypes = [i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]; l.rodent_get_film_data.restype = None
    l.hip_generate_rays.argtypes = [i32, C.POINTER(PrimaryStream), i32, i32, i32, C.POINTER(Settings), i32, i32, i32, i32, i32,
        vp]; l.hip_generate_rays.restype = None
    l.hip_traverse_primary.argtypes = [i32, C.POINTER(PrimaryStream), vp]; l.hip_traverse_primary.restype = None
    l.hip_sort_primary.argtypes = [i32, C.POINTER(PrimaryStream), C.POINTER(PrimaryStream), C.POINTER(i32),
        vp]; l.hip_sort_primary.restype = None
    l.hip_shade.argtypes = [i32, C.POINTER(PrimaryStream), C.POINTER(SecondaryStream), i32, vp]; l.hip_shade.restype = None
    l.hip_traverse_secondary.argtypes = [i32, C.POINTER(SecondaryStream), vp]; l.hip_traverse_secondary.restype = None
    l.hip_compact_primary.argtypes = [i32, C.POINTER(PrimaryStream), C.POINTER(PrimaryStream), vp]; l.hip_compact_primary.restype = i32
    return l


def write_stream_array(ptr, values):
    """Copies a host array of 4-byte words into one stream array (device pointer)."""
    import torch
    a = np.ascontiguousarray(values)
    assert a.dtype.itemsize == 4
    if a.size == 0:
        return
    t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()
    abi.lib()
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(ptr), C.c_void_p(t.data_ptr()), C.c_size_t(a.nbytes), 3)
    torch.cuda.synchronize()


def read_stream_array(ptr, count, dtype):
    """Copies `count` words of one stream array (device pointer) to the host."""
    import torch
    if count == 0:
        return np.zeros(0, dtype)
    nbytes = count * 4
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    abi.lib()  # same process / same HIP runtime
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3)
    return t.cpu().numpy().view(dtype).copy()
