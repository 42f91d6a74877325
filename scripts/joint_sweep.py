#!/usr/bin/env python
"""Where the joint traversal launch (rodent_hip_render_trace_persistent(dev, 2): shadow pass of an iteration inside the next
iteration's closest-hit launch, persistent 16-wave workgroups with a 255-node image, one stream) beats the default (2-wave
workgroups with a 31-node image, shadow pass on a second stream): the atrium decimated to 1/1 ... 1/512 of its faces and the
Cornell box, streaming mapping, 1920 x 1080 x 16 spp, path length 8.  usage: python scripts/joint_sweep.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import render as R, scene as S, scenes

scenes.scene_bvh("atrium")
src = scenes.DATA / "atrium.obj"
mtl = (scenes.DATA / "atrium.mtl").read_text()
emissive, cur = set(), None
for line in mtl.splitlines():
    t = line.split()
    if t[:1] == ["newmtl"]:
        cur = t[1]
    if t[:1] == ["Ke"] and any(float(x) > 0 for x in t[1:4]):
        emissive.add(cur)


def decimate(keep_every, dst):
    k, mat = 0, None
    with open(src) as f, open(dst, "w") as out:
        for line in f:
            if line.startswith("usemtl"):
                mat = line.split()[1]
            if line.startswith("f "):
                if mat in emissive or k % keep_every == 0:
                    out.write(line)
                k += 1
            else:
                out.write(line)
    return dst


W, H, SPP, LEN = 1920, 1080, 16, 8
cases = [("cornell", scenes.GOLDEN / "cornell_box.obj", scenes.CAMERAS["cornell"])]
for k in (512, 128, 32, 8, 2, 1):
    dst = Path("/tmp") / f"atrium-keep{k}.obj"
    if k > 1:
        (Path("/tmp") / "atrium.mtl").write_text(mtl)
        decimate(k, dst)
    cases.append((f"atrium 1/{k}", dst if k > 1 else src, scenes.CAMERAS["atrium"]))
print(f"{'scene':14s} {'BVH nodes':>9s} {'default':>9s} {'joint':>9s}   Msamples/s, streaming mapping, {W}x{H}x{SPP} spp, path length {LEN}")
for name, obj, (eye, d, up, fov) in cases:
    sc = S.convert(obj, Path("/tmp") / "sweep.rscene")
    cam = S.camera_settings(eye, d, up, fov, W, H)
    rates = {}
    for mode in (0, 2):
        r = R.Renderer(sc, W, H, SPP, LEN, mapping="streaming", trace_persistent=mode)
        r.render(cam, 0); secs = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_rows(cam, it + 1, 0, H); secs.append(time.perf_counter() - t0)
        rates[mode] = SPP * W * H / float(np.median(secs)) / 1e6
        r.close()
    print(f"{name:14s} {len(sc.nodes):9d} {rates[0]:9.1f} {rates[2]:9.1f}   joint / default = {rates[2] / rates[0]:.3f}", flush=True)
