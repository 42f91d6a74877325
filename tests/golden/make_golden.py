"""Generates the committed golden fixtures in tests/golden/.

Run from the repo root:  python tests/golden/make_golden.py

Inputs : tests/golden/cornell_box.obj/.mtl  (data files of the reference's own test
         suite, /root/reference/testing/cornell_box.{obj,mtl}; public domain).
Outputs: cornell.bvh                  BVH8 + BVH4 + BVH2 blocks (rodent_amd/bin/bvh_extractor)
         cornell-primary-64x64.rays   64x64 pinhole rays, eye 0 1 2.7, dir 0 0 -1, fov 60
         cornell-random-4096.rays     4096 random segments in the scene bounds, seed 42
         cornell-edge.rays            hand-made edge cases (see edge_rays())
         cornell-expected.npz         Hit1 arrays from the CPU oracle for every
                                      (algorithm, ray set, closest/any) combination

The reference itself cannot produce these vectors here (its traversal is Impala and
needs AnyDSL), so they are pinned by the oracle and cross-checked by the exhaustive
checker in tests/test_oracle.py ("parity unpinned by the reference").
"""
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import binding as O            # noqa: E402
from rodent_amd import build, formats as F  # noqa: E402

G = ROOT / "tests" / "golden"

RAY_SETS = {  # name -> (file, tmin, tmax)   (flags of README.md:34-37 / cmake/test/run_traversal.cmake)
    "primary": ("cornell-primary-64x64.rays", 0.0, 5000.0),
    "primary_tmin": ("cornell-primary-64x64.rays", 0.01, 5000.0),
    "random": ("cornell-random-4096.rays", 0.0, 1.0),
    "edge": ("cornell-edge.rays", 0.0, 100.0),
}
ALGOS = {  # name -> (bvh block, width, oracle algo)
    "bvh2_gpu": (F.BVH2_TRI1, 2, "ref"),    # mapping_gpu.impala:94-178 on BVH2/Tri1
    "bvh4_cpu": (F.BVH4_TRI4, 4, "ref"),    # mapping_cpu.impala:138-256 on BVH4/Tri4
    "bvh8_cpu": (F.BVH8_TRI4, 8, "ref"),    # mapping_cpu.impala:138-256 on BVH8/Tri4
    "bvh4_gpu": (F.BVH4_TRI4, 4, "gpu"),    # mapping_gpu.impala general-arity branch (:136-153) on BVH4/Tri4
    "bvh8_gpu": (F.BVH8_TRI4, 8, "gpu"),    # mapping_gpu.impala general-arity branch on BVH8/Tri4
}


def edge_rays():
    o, d = [], []
    # axis-parallel rays (a zero direction component -> safe_rcp clamps to +-FLT_MAX)
    for x in np.linspace(-0.9, 0.9, 7):
        for y in np.linspace(0.1, 1.9, 7):
            o.append((x, y, 2.0)); d.append((0.0, 0.0, -1.0))
            o.append((x, 3.0, y - 1.0)); d.append((0.0, -1.0, 0.0))
            o.append((-3.0, y, x)); d.append((1.0, 0.0, 0.0))
    # rays through exact vertices / edges of the scene (ties between adjacent triangles)
    for p in [(-1.01, 0.0, 0.99), (1.0, 0.0, 0.99), (1.0, 0.0, -1.04), (-0.99, 0.0, -1.04), (0.0, 0.0, 0.0),
              (-0.24, 1.98, 0.16), (0.23, 1.98, -0.22)]:
        o.append((0.0, 1.0, 2.7)); d.append(tuple(np.float32(p) - np.float32((0.0, 1.0, 2.7))))
    # origin exactly on a surface, pointing in / out / along it
    for dd in [(0, 1, 0), (0, -1, 0), (1, 0, 0), (0.3, 1e-9, 0.2)]:
        o.append((0.1, 0.0, 0.1)); d.append(dd)
    # degenerate directions: zero, denormal-small, huge
    o.append((0.0, 1.0, 0.0)); d.append((0.0, 0.0, 0.0))
    o.append((0.0, 1.0, 0.0)); d.append((1e-30, 1e-30, -1e-30))
    o.append((0.0, 1.0, 2.0)); d.append((0.0, 0.0, -1e20))
    # rays that start outside and miss everything
    for k in range(16):
        o.append((5.0 + k, 5.0, 5.0)); d.append((1.0, 0.5 * k, 0.25))
    # unnormalised short segments ending just before / after a wall (t close to 1)
    for s in (0.5, 0.999, 1.0, 1.001, 2.0):
        o.append((0.0, 1.0, 0.0)); d.append((0.0, 0.0, -1.04 * s))
    return F.make_rays(np.array(o, "<f4"), np.array(d, "<f4"))


def main():
    build.build_host()
    tools = build.BIN_DIR
    subprocess.run([tools / "bvh_extractor", "-obj", G / "cornell_box.obj", "-o", G / "cornell.bvh"], check=True)
    subprocess.run([tools / "ray_gen", "primary", "0", "1", "2.7", "0", "0", "-1", "0", "1", "0", "60", "64", "64",
                    G / "cornell-primary-64x64.rays"], check=True)
    subprocess.run([tools / "ray_gen", "random", G / "cornell.bvh", "4096", "42", G / "cornell-random-4096.rays"], check=True)
    F.write_rays(G / "cornell-edge.rays", edge_rays())

    out = {}
    for aname, (block, width, algo) in ALGOS.items():
        nodes, tris = F.read_bvh(G / "cornell.bvh", block)
        for rname, (rf, tmin, tmax) in RAY_SETS.items():
            rays = F.read_rays(G / rf, tmin, tmax)
            for any_hit in (False, True):
                hits, _ = O.traverse(width, nodes, tris, rays, any_hit=any_hit, algo=algo)
                out[f"{aname}.{rname}.{'any' if any_hit else 'closest'}"] = hits
    np.savez_compressed(G / "cornell-expected.npz", **out)
    print("wrote", len(out), "expected hit arrays")


if __name__ == "__main__":
    main()
