import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rodent_amd import abi, formats as F, raygen, scenes
path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
n4, _ = F.read_bvh(path, F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(n4)
for name, rays in (("primary", raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0)),
    ("random", raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0))):
    n = len(rays); rd = abi.to_device(rays, 0); hd = torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0")
    for any_hit in (False, True):
        abi.read_stats(0)
        abi.traverse_async(bvh, rd, hd, n, any_hit, 0, torch.cuda.current_stream()); torch.cuda.synchronize()
        print(name, any_hit, abi.read_stats(0))
