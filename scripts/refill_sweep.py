#!/usr/bin/env python
"""Lane refill in the renderer's persistent traversal launches (rodent_hip_render_trace_refill): the atrium (and two decimated
forms) and the Cornell box through the streaming mapping with the joint launch, whole chunks (0) against refill once 16 / 24 / 32 / 48
lanes of a wave are idle.  Ray counts must agree exactly, films up to the order of the atomic adds.
usage: python scripts/refill_sweep.py [--frames 3] [--size 1920x1080] [--spp 16]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import render as R, scene as S, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--idle", default="0,32,48,48:32,32:48", help="thresholds, n or bounce:shadow")
ap.add_argument("--scenes", default="atrium,atrium/8,atrium/128,cornell", help="atrium/K = the atrium with every K-th face")
a = ap.parse_args()
W, H = (int(x) for x in a.size.split("x"))
SPP, LEN = a.spp, 8
scenes.scene_bvh("atrium")
cases = {"atrium": (scenes.DATA / "atrium.obj", scenes.CAMERAS["atrium"]),
    "cornell": (scenes.GOLDEN / "cornell_box.obj", scenes.CAMERAS["cornell"])}


def decimated(keep_every):
    """The atrium with every keep_every-th face (and every emissive one), as scripts/joint_sweep.py builds it."""
    src, mtl = scenes.DATA / "atrium.obj", (scenes.DATA / "atrium.mtl").read_text()
    emissive, cur = set(), None
    for line in mtl.splitlines():
        t = line.split()
        if t[:1] == ["newmtl"]:
            cur = t[1]
        if t[:1] == ["Ke"] and any(float(x) > 0 for x in t[1:4]):
            emissive.add(cur)
    dst = Path("/tmp") / f"atrium-keep{keep_every}.obj"
    (Path("/tmp") / "atrium.mtl").write_text(mtl)
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


for name in a.scenes.split(","):
    if name.startswith("atrium/"):
        cases[name] = (decimated(int(name.split("/")[1])), scenes.CAMERAS["atrium"])
idles = [tuple(int(y) for y in x.split(":")) if ":" in x else int(x) for x in a.idle.split(",")]
print(f"{'scene':10s} " + " ".join(f"{'idle ' + str(i).replace(' ', ''):>12s}"
    for i in idles) + f"   Msamples/s, streaming mapping, joint persistent launch, {W}x{H}x{SPP} spp, path length {LEN}")
for name in a.scenes.split(","):
    obj, (eye, d, up, fov) = cases[name]
    sc = S.convert(obj, Path("/tmp") / "refill.rscene")
    cam = S.camera_settings(eye, d, up, fov, W, H)
    rates, ref_film, ref_counts, notes = [], None, None, []
    for idle in idles:
        r = R.Renderer(sc, W, H, SPP, LEN, mapping="streaming", trace_persistent=2, trace_refill=idle)
        r.render(cam, 0)
        film = r.film().copy(); counts = r.counters()
        if ref_film is None:
            ref_film, ref_counts = film, counts
        else:
            same_counts = all(counts[k] == ref_counts[k] for k in ("primary_rays", "shadow_rays"))
            err = float(np.abs(film - ref_film).max() / max(float(np.abs(ref_film).max()), 1e-30))
            notes.append(f"idle {idle}: ray counts {'equal' if same_counts else 'DIFFER ' + str(counts)}, film max rel diff {err:.1e}")
        secs = []
        for it in range(a.frames):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_rows(cam, it + 1, 0, H); secs.append(time.perf_counter() - t0)
        rates.append(SPP * W * H / float(np.median(secs)) / 1e6)
        r.close()
    print(f"{name:10s} {len(sc.nodes):7d} nodes " + " ".join(f"{x:12.1f}" for x in rates) + "   " + "; ".join(notes), flush=True)
