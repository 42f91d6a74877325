#!/usr/bin/env python
"""What LDS-staged top-of-tree nodes could take off the vector-memory path: the share of B1's inner-node visits that falls on the top D
levels of the hierarchy (2^D - 1 nodes at most), per ray set, from the oracle's per-node visit counts (oracle.binding.node_visits). usage:
python scripts/model_top_levels.py data/atrium.bvh data/atrium-primary.rays [data/atrium-random.rays ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import binding as O
from rodent_amd import formats as F

nodes, tris = F.read_bvh(sys.argv[1], F.BVH2_TRI1)
child = np.asarray(nodes["child"]).reshape(-1, 2)
depth = np.full(len(nodes), -1, np.int64)
depth[0] = 0
frontier = [0]
while frontier:                                               # breadth-first: child ids > 0 are inner nodes (id - 1)
    nxt = []
    for i in frontier:
        for c in child[i]:
            if c > 0:
                depth[c - 1] = depth[i] + 1
                nxt.append(c - 1)
    frontier = nxt
print(f"{len(nodes)} nodes, depth up to {depth.max()}")
for path in sys.argv[2:]:
    rays = F.read_rays(path, 0.0, 5000.0 if "primary" in path else 1.0)
    visits = O.node_visits(nodes, tris, rays).astype(np.float64)
    total = visits.sum()
    print(f"{path}: {total / len(rays):.1f} inner-node visits per ray")
    for d in range(1, 13):
        top = depth < d
        print(f"  top {d:2d} levels ({int(top.sum()):5d} nodes, {int(top.sum()) * 64 / 1024:7.1f} KB): {visits[top].sum() / total:6.1%} of "
            f"the visits")
