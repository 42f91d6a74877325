#!/usr/bin/env python
"""CPU frame rate of the reference's CPU mapping, printed like `rodent --bench` (driver.cpp:344-347).

TEST INFRASTRUCTURE: a CPU Msamples/s figure to quote beside the GPU renderer's.
  --mapping wavefront (default): the reference's tile-parallel wavefront renderer restated (oracle/cpu_wavefront.inc:
      render/mapping_cpu.impala:352-473 -- 16x16 tiles from an atomic counter, per-thread primary / secondary streams,
      generate -> hybrid ray8 x BVH8 traversal -> sort by geometry -> shade -> compact -> shadow rays), scalar shading;
  --mapping scalar: the parity oracle, one path at a time over the BVH2 single-ray kernel, rows split over threads.
usage: python oracle/cpu_render_bench.py [--scene file.obj|file.rscene] [--width 1920 --height 1080 --spp 64
       --max-path-len 4 --bench 2 --threads N --mapping wavefront|scalar]"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import binding as O                 # noqa: E402
from rodent_amd import scene as S               # noqa: E402

ROOT = Path(__file__).resolve().parents[1]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=str(ROOT / "tests" / "golden" / "cornell_box.obj"))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--max-path-len", type=int, default=4)
    ap.add_argument("--bench", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--eye", type=float, nargs=3, default=[0, 1, 2.7])
    ap.add_argument("--dir", type=float, nargs=3, default=[0, 0, -1])
    ap.add_argument("--up", type=float, nargs=3, default=[0, 1, 0])
    ap.add_argument("--fov", type=float, default=60.0)
    ap.add_argument("--mapping", choices=("wavefront", "scalar"), default="wavefront")
    a = ap.parse_args(argv)
    path = Path(a.scene)
    scene = S.Scene(path) if path.suffix == ".rscene" else S.convert(path, Path("/tmp") / (path.stem + ".cpu_bench.rscene"))
    cam = S.camera_settings(a.eye, a.dir, a.up, a.fov, a.width, a.height)
    if a.mapping == "wavefront":
        # the reference's CPU targets trace a BVH8 / Tri4 (converter.cpp:152-259): built from the scene's own triangles
        import subprocess
        from rodent_amd import build, formats as F
        if path.suffix == ".rscene":
            raise SystemExit("--mapping wavefront needs the .obj (it builds the BVH8 the reference's CPU mapping traces)")
        bvh = Path("/tmp") / (path.stem + ".cpu_bench.bvh")
        subprocess.run([str(build.BIN_DIR / "bvh_extractor"), "-obj", str(path), "-o", str(bvh)], check=True, stdout=subprocess.DEVNULL)
        n8, t8 = F.read_bvh(bvh, F.BVH8_TRI4)
    film, rates, rays = None, [], 0
    for it in range(a.bench):
        t0 = time.perf_counter()
        if a.mapping == "wavefront":
            film, counts = O.render_wavefront(scene, n8, t8, cam, it, a.spp, a.max_path_len, a.width, a.height, film, threads=a.threads)
        else:
            film, counts = O.render(scene, cam, it, a.spp, a.max_path_len, a.width, a.height, film, threads=a.threads)
        dt = time.perf_counter() - t0
        rates.append(a.spp * a.width * a.height / dt / 1e6)
        rays += int(counts.sum())
    rates.sort()
    print(f"# {rates[0]:g}/{rates[len(rates) // 2]:g}/{rates[-1]:g} (min/med/max Msamples/s)  [{a.mapping} mapping, {a.threads} threads, "
        f"{rays} rays]")
    assert np.isfinite(film).all()
    return 0


if __name__ == "__main__":
    sys.exit(main())
