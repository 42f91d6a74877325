#!/usr/bin/env python
"""How deep the traversal stack of the benchmark rays gets (oracle.binding.ray_depths): the share of rays an LDS window of N
entries cannot hold -- those restart in the one-wave follow-up kernel -- per ray set.
usage: python scripts/model_stack_depth.py data/atrium.bvh data/atrium-primary.rays data/atrium-random.rays"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import binding as O
from rodent_amd import formats as F

nodes, tris = F.read_bvh(sys.argv[1], F.BVH2_TRI1)
for path in sys.argv[2:]:
    rays = F.read_rays(path, 0.0, 5000.0 if "primary" in path else 1.0)
    depth = O.ray_depths(nodes, tris, rays)
    print(f"{path}: deepest stack pointer mean {depth.mean():.2f}, max {depth.max()}")
    for window in (8, 10, 11, 12, 13, 14, 15, 16):
        over = int((depth >= window).sum())
        print(f"  window of {window:2d} entries: {over:7d} rays ({over / len(rays):8.4%}) do not fit")
