"""Ray dump generation in numpy, bit-identical to the host tool rodent_amd/bin/ray_gen
(which restates tools/ray_gen/ray_gen.cpp:20-58 primary, :87-111 random).

Used where a process needs its own slice of a larger ray set without writing files:
multi-GPU runs give rank r sub-pixel sample r of N (primary) or seed 42 + r (random).
"""
from __future__ import annotations

import numpy as np

from . import formats as F

f32 = np.float32


def _normalize(v):
    v = np.asarray(v, f32)
    l = np.sqrt(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=f32)   # same order as dot(): x*x + y*y + z*z
    return v * (f32(1.0) / l)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], f32)


def primary_rays(eye, direction, up, fov, width, height, tmin=0.0, tmax=1e9, sample=0, num_samples=1):
    """Pinhole rays, rows top to bottom, UNNORMALISED directions (ray_gen.cpp:41-52).

    sample/num_samples: sub-pixel offset (sample + 0.5) / num_samples inside each pixel; the
    default (0, 1) is the pixel centre, i.e. exactly what the reference tool writes."""
    eye = np.asarray(eye, f32); d_in = np.asarray(direction, f32); up_in = np.asarray(up, f32)
    d = _normalize(d_in)
    right = _normalize(_cross(d_in, up_in))
    upv = _normalize(_cross(right, d_in))
    scale = f32(np.tan(np.float64(f32(fov)) * (np.pi / 360.0)))   # the tool evaluates tan in double, then rounds
    right = right * scale
    upv = upv * (f32(height) / f32(width) * scale)
    sx, sy = f32(2.0) / f32(width), f32(2.0) / f32(height)
    off = f32((sample + 0.5) / num_samples)
    j = np.arange(width, dtype=f32); i = np.arange(height - 1, -1, -1, dtype=f32)
    kx = sx * (j + off) - f32(1.0)
    ky = sy * (i + off) - f32(1.0)
    dirs = (d[None, None, :] + kx[None, :, None] * right[None, None, :]) + ky[:, None, None] * upv[None, None, :]
    org = np.broadcast_to(eye, dirs.shape)
    return F.make_rays(org.reshape(-1, 3), dirs.reshape(-1, 3).astype(f32), tmin, tmax)


def _splitmix_uniform(seed, count):
    """count floats in [0,1) from splitmix64 started at `seed` (ray_gen.cpp SplitMix::uni)."""
    gamma = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + gamma * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(f32) * f32(1.0 / 16777216.0)


def scene_bounds(nodes4):
    """Union of the BVH4 root's child boxes (ray_gen.cpp:134-144)."""
    b = nodes4["bounds"][0]
    return b[[0, 2, 4]].min(axis=1).astype(f32), b[[1, 3, 5]].max(axis=1).astype(f32)


def scene_bounds2(nodes2):
    """The same box from a BVH2 block (a .bvh without a BVH4 block: the generated scenes): union of the root's two child boxes."""
    b = nodes2["bounds"][0].reshape(2, 6)
    b = b[np.isfinite(b).all(axis=1)]
    return b[:, [0, 2, 4]].min(axis=0).astype(f32), b[:, [1, 3, 5]].max(axis=0).astype(f32)


def shadow_rays(light, rays, t, tmin=0.0, tmax=1.0):
    """ray_gen's third mode (tools/ray_gen/ray_gen.cpp:60-85): from a point light towards the hit points of a previous pass -- org = light,
    dir = (org + t * dir) - light, t = the .fbuf of that pass (the miss value where a ray hit nothing).  Traced any-hit with tmax just under
    1 this is the suite's occlusion class (benchmarks/benchmark.py:36-41 "ao": -any, short tmax)."""
    light = np.asarray(light, f32)
    hit = rays["org"] + np.asarray(t, f32)[:, None] * rays["dir"]
    return F.make_rays(np.broadcast_to(light, hit.shape), (hit - light).astype(f32), tmin, tmax)


def random_rays(lo, hi, count, seed, tmin=0.0, tmax=1.0):
    """Segments between two uniform points of the box: org = p1, dir = p2 - p1 (ray_gen.cpp:97-103)."""
    lo = np.asarray(lo, f32); hi = np.asarray(hi, f32)
    u = _splitmix_uniform(seed, 6 * count).reshape(count, 6)
    ext = hi - lo
    p1 = lo + ext * u[:, :3]
    p2 = lo + ext * u[:, 3:]
    return F.make_rays(p1, p2 - p1, tmin, tmax)
