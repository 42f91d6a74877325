"""Regenerable benchmark inputs (scene -> .bvh -> ray dumps), built with the host tools.

BASELINE.json's inputs (testing/sponza.bvh, sponza-primary.rays, sponza-random.rays)
are absent from the reference checkout, so the workloads are:
  * "sponza"  -- used as-is if data/sponza.bvh + data/sponza-{primary,random}.rays exist;
  * "atrium"  -- seeded procedural Sponza-class scene (host/atrium.cpp), ~265 K triangles;
  * "cornell" -- the reference's testing/cornell_box.obj (36 triangles);
  * "gallery", "crown", "plant" -- the other scene classes of the reference's benchmark suite (benchmarks/benchmark.py:16-21) as seeded
    stand-ins (host/atrium.cpp at detail 4, host/stress_scenes.cpp): 4.2 M-triangle architecture, a 4.2 M-triangle organic surface,
    2.1 M long thin triangles; traversal only (a .bvh with a BVH2 block, built where it is needed; "<scene>/<detail>" = a smaller build).
Ray dumps follow SURVEY.md 8(d): 1024x1024 primary rays (fov 60, unnormalised
directions; README.md:34-37 uses --tmax 5000) and 1 Mi random segments (--tmax 1), seed 42.
"""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

from . import build

ROOT = Path(__file__).resolve().parent.parent
DATA = Path(os.environ.get("RODENT_DATA_DIR", ROOT / "data"))
GOLDEN = ROOT / "tests" / "golden"

# camera used for the primary dumps: (eye, dir, up, fov)
CAMERAS = {
    "atrium": ((-1150.0, 350.0, 30.0), (1.0, 0.12, -0.05), (0.0, 1.0, 0.0), 60.0),
    "cornell": ((0.0, 1.0, 2.7), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 60.0),
    "gallery": ((-1150.0, 350.0, 30.0), (1.0, 0.12, -0.05), (0.0, 1.0, 0.0), 60.0),
    "crown": ((150.0, 420.0, 1500.0), (-0.08, -0.22, -1.0), (0.0, 1.0, 0.0), 50.0),
    "plant": ((-1900.0, 760.0, -1150.0), (1.0, -0.18, 0.62), (0.0, 1.0, 0.0), 65.0),
}
# point lights of the "ao" ray class (ray_gen shadow, tools/ray_gen/ray_gen.cpp:60-85): rays from the light to the camera rays' hit points
LIGHTS = {"atrium": (0.0, 1450.0, 0.0), "gallery": (0.0, 1450.0, 0.0), "crown": (600.0, 1500.0, 900.0), "plant": (0.0, 1350.0, 0.0),
    "cornell": (0.0, 1.9, 0.0)}
GENERATED = {"gallery": 4, "crown": 4, "plant": 4}          # scene_gen kinds built straight to a .bvh, with their default detail
PRIMARY_TMAX, RANDOM_TMAX = 5000.0, 1.0


def _tool(name) -> Path:
    p = build.BIN_DIR / name
    if not p.exists():
        build.build_host()
    return p


def _run(cmd):
    subprocess.run([str(c) for c in cmd], check=True, stdout=subprocess.DEVNULL)


def scene_bvh(scene: str) -> Path:
    DATA.mkdir(parents=True, exist_ok=True)
    kind, _, detail = scene.partition("/")
    if kind in GENERATED:                                   # "plant", "plant/1": BVH2 block only, no OBJ round trip
        detail = int(detail) if detail else GENERATED[kind]
        out = DATA / f"{kind}-d{detail}.bvh"
        if not out.exists():
            _run([_tool("scene_gen"), kind, "--bvh", out, 1, detail])
        return out
    out = DATA / f"{scene}.bvh"
    if out.exists():
        return out
    if scene == "atrium":
        obj = DATA / "atrium.obj"
        if not obj.exists():
            _run([_tool("scene_gen"), "atrium", obj, 1])
    elif scene == "cornell":
        obj = GOLDEN / "cornell_box.obj"
    else:
        raise FileNotFoundError(f"{out} not found and scene '{scene}' cannot be generated")
    _run([_tool("bvh_extractor"), "-obj", obj, "-o", out])
    return out


def scene_obj(scene: str) -> Path:
    """OBJ (+ atrium.mtl beside it) of a scene, for the renderer's converter: the atrium, or a generated kind ("gallery", "crown/2",
    ...)."""
    DATA.mkdir(parents=True, exist_ok=True)
    kind, _, detail = scene.partition("/")
    if kind == "cornell":
        return GOLDEN / "cornell_box.obj"
    if kind == "atrium":
        scene_bvh("atrium")
        return DATA / "atrium.obj"
    detail = int(detail) if detail else GENERATED[kind]
    out = DATA / f"{kind}-d{detail}.obj"
    if not out.exists():
        _run([_tool("scene_gen"), kind, out, 1, detail])
        if kind in PANELS:                                  # the stress scenes are geometry only: the renderer needs something that emits
            _append_panels(out, PANELS[kind])
    return out


# Emissive panels ("light" of atrium.mtl, facing down) for the renderer's frames of the generated stress scenes: (centre x, y, z, half
# size). Appended to the OBJ only -- the traversal matrix builds its .bvh straight from the generator (scene_bvh) and does not see them.
PANELS = {"crown": [(0.0, 950.0, 0.0, 450.0)],
          "plant": [(x, 1390.0, z, 150.0) for x in (-1300.0, 0.0, 1300.0) for z in (-600.0, 600.0)]}


def _append_panels(obj: Path, panels):
    count = 0
    with open(obj) as f:
        for line in f:
            count += line.startswith("v ")
    with open(obj, "a") as f:
        f.write("usemtl light\n")
        for cx, y, cz, r in panels:
            for x, z in ((cx - r, cz - r), (cx + r, cz - r), (cx + r, cz + r), (cx - r, cz + r)):      # (b - a) x (c - a) points down
                f.write(f"v {x} {y} {z}\n")
            f.write(f"f {count + 1} {count + 2} {count + 3}\nf {count + 1} {count + 3} {count + 4}\n")
            count += 4


def primary_rays(scene: str, width=1024, height=1024) -> Path:
    sfx = "" if (width, height) == (1024, 1024) else f"-{width}x{height}"
    out = DATA / f"{scene}-primary{sfx}.rays"
    if not out.exists():
        scene_bvh(scene)
        eye, d, up, fov = CAMERAS[scene]
        _run([_tool("ray_gen"), "primary", *eye, *d, *up, fov, width, height, out])
    return out


def random_rays(scene: str, count=1 << 20, seed=42) -> Path:
    sfx = "" if count == 1 << 20 else f"-{count}"
    out = DATA / f"{scene}-random{sfx}.rays"
    if not out.exists():
        _run([_tool("ray_gen"), "random", scene_bvh(scene), count, seed, out])
    return out


def default_scene() -> str:
    """'sponza' when the real blobs were dropped into data/, else the atrium."""
    if all((DATA / f).exists() for f in ("sponza.bvh", "sponza-primary.rays", "sponza-random.rays")):
        return "sponza"
    return "atrium"
