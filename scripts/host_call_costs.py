#!/usr/bin/env python
"""What the host pays around a launch of the reference-named synchronous entry points: the HIP runtime calls
they make, timed one by one on this box (ctypes on libamdhip64: no Python work inside the timed loops beyond the call itself),
then the entry point itself: wall time per call against the kernel time it accounts (rodent_hip_get_kernel_time).
usage: python scripts/host_call_costs.py [--calls 2000]"""
import argparse, ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=2000); a = ap.parse_args()
hip = C.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda:0")
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")
ptr = C.c_void_p(buf.data_ptr() + 4096)

def per_call(fn, n=a.calls):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6

base, size, dev = C.c_void_p(), C.c_size_t(), C.c_int()
ev = [C.c_void_p(), C.c_void_p()]
hip.hipEventCreate(C.byref(ev[0])); hip.hipEventCreate(C.byref(ev[1]))
pinned = C.c_void_p(); hip.hipHostMalloc(C.byref(pinned), 128, 0)
ms = C.c_float()
rows = [("python ctypes call overhead (hipGetDevice)", lambda: hip.hipGetDevice(C.byref(dev))),
        ("hipSetDevice(0) (already current)", lambda: hip.hipSetDevice(0)),
        ("hipMemGetAddressRange", lambda: hip.hipMemGetAddressRange(C.byref(base), C.byref(size), ptr)),
        ("hipEventRecord (null stream)", lambda: hip.hipEventRecord(ev[0], None)),
        ("hipStreamSynchronize (idle null stream)", lambda: hip.hipStreamSynchronize(None)),
        ("hipMemcpyAsync D2H 80 B into pinned + hipStreamSynchronize",
            lambda: (hip.hipMemcpyAsync(pinned, ptr, 80, 2, None), hip.hipStreamSynchronize(None))),
        ("hipEventRecord x 2 + hipStreamSynchronize + hipEventElapsedTime",
            lambda: (hip.hipEventRecord(ev[0], None), hip.hipEventRecord(ev[1], None), hip.hipStreamSynchronize(None),
            hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])))]
for name, fn in rows:
    print(f"{per_call(fn):8.2f} us  {name}")

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
rays = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0)
rd = abi.to_device(rays, 0); hd = torch.zeros(len(rays) * 16, dtype=torch.uint8, device="cuda:0")
l = abi.lib()
call = lambda: l.amdgpu_intersect_single_ray1_bvh2_tri1(0, bvh.nodes.data_ptr(), bvh.tris.data_ptr(), rd.data_ptr(), hd.data_ptr(),
    len(rays))
for _ in range(20): call()
k0 = l.rodent_hip_get_kernel_time(); t0 = time.perf_counter()
n = 200
for _ in range(n): call()
wall = (time.perf_counter() - t0) / n * 1e6; kern = (l.rodent_hip_get_kernel_time() - k0) / n
print(f"amdgpu_intersect_single_ray1_bvh2_tri1, 1 Mi primary rays: {wall:.2f} us wall per call, {kern:.2f} us of kernels by the library's "
    f"account -> {wall - kern:.2f} us of host")
st = torch.cuda.current_stream()
for _ in range(20): abi.traverse_async(bvh, rd, hd, len(rays), False, 0, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): abi.traverse_async(bvh, rd, hd, len(rays), False, 0, st)
torch.cuda.synchronize()
print(f"hip_traverse_bvh2_tri1_async back to back on one stream: {(time.perf_counter() - t0) / n * 1e6:.2f} us per launch")
