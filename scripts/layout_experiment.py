#!/usr/bin/env python
"""Node ORDER experiment: the same BVH2 with its Node2 records permuted in memory (child indices remapped, root kept at 0,
leaves untouched), traced by the default kernel.  A Node2 is 64 bytes and an L2 line is 128, so the order decides which
second node a miss brings along:

  file        the builder's order (depth-first preorder: the first child follows its parent)
  bfs         breadth-first
  pairs-dfs   the two inner children of a node share an aligned 128-byte line; pairs in depth-first order
  pairs-bfs   the same, pairs in breadth-first order
  shifted     file order moved by one slot behind the root (flips which neighbours share a line)

Hits are identical by construction (checked).  usage: python scripts/layout_experiment.py [--steps 30] [--variant top]"""
import argparse
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, formats as F, raygen, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--variant", default="top")
ap.add_argument("--scene", default="atrium")
a = ap.parse_args()

nodes, tris = F.read_bvh(scenes.scene_bvh(a.scene), F.BVH2_TRI1)
n_nodes = len(nodes)
child = nodes["child"]


def inner_children(i):
    return [int(c) - 1 for c in child[i] if c > 0]          # an inner child is stored as index + 1


def order_bfs():
    out, queue = [], [0]
    while queue:
        nxt = []
        for i in queue:
            out.append(i)
            nxt += inner_children(i)
        queue = nxt
    return out


def order_pairs(depth_first):
    """Slots: root at 0, a filler at 1, then sibling groups; a group of two starts on an even slot (fillers = -1)."""
    out = [0, -1]
    work = [inner_children(0)]
    while work:
        group = work.pop() if depth_first else work.pop(0)
        if not group:
            continue
        if len(group) == 2 and len(out) % 2:
            out.append(-1)
        out += group
        kids = [inner_children(g) for g in group]
        if depth_first:
            work += reversed(kids)
        else:
            work += kids
    return out


def order_shifted():
    return [0, -1] + list(range(1, n_nodes))


def apply(order):
    """order[slot] = old node index (or -1 = filler: a copy of the root, never referenced)."""
    order = np.asarray(order)
    new_of_old = np.zeros(n_nodes, np.int64)
    real = order >= 0
    new_of_old[order[real]] = np.nonzero(real)[0]
    out = nodes[np.where(real, order, 0)].copy()
    c = out["child"]
    out["child"] = np.where(c > 0, new_of_old[np.maximum(c - 1, 0)] + 1, c).astype(np.int32)
    assert real.sum() == n_nodes and len(set(order[real].tolist())) == n_nodes
    return out


def share_stats(nd):
    """Fraction of inner-child links whose target lies in the parent's 128-byte line / whose two inner children share one."""
    inner = nd["child"] > 0
    c = nd["child"] - 1
    idx = np.arange(len(nd))[:, None]
    with_parent = ((c >> 1) == (idx >> 1)) & inner
    both = inner.all(1)
    sib = both & ((c[:, 0] >> 1) == (c[:, 1] >> 1))
    return with_parent.sum() / max(inner.sum(), 1), sib.sum() / max(both.sum(), 1)


eye, d, up, fov = scenes.CAMERAS[a.scene]
prim = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX)
nodes4, _ = F.read_bvh(scenes.scene_bvh(a.scene), F.BVH4_TRI4)
lo, hi = raygen.scene_bounds(nodes4)
rnd = raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, scenes.RANDOM_TMAX)
names = abi.variants(2)
v = names.index(a.variant)


def timed(bvh, rd, hd, n, steps):
    st = torch.cuda.current_stream()
    for _ in range(5):
        abi.traverse_async(bvh, rd, hd, n, False, v, st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(st); abi.traverse_async(bvh, rd, hd, n, False, v, st); e.record(st)
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev]))


sets = {"primary": prim, "random": rnd}
dev = {k: (abi.to_device(r, 0), torch.zeros(len(r) * 16, dtype=torch.uint8, device="cuda:0")) for k, r in sets.items()}
ref = {}
print(f"{a.scene}: {n_nodes} Node2 records, variant {a.variant}")
for label, order in (("file", list(range(n_nodes))), ("shifted", order_shifted()), ("bfs", order_bfs()),
                     ("pairs-dfs", order_pairs(True)), ("pairs-bfs", order_pairs(False))):
    nd = apply(order)
    wp, sib = share_stats(nd)
    bvh = abi.DeviceBvh(2, nd, tris, 0)
    line = f"  {label:10s} {len(nd):7d} slots; child in its parent's line {wp:.2f}, sibling pairs in one line {sib:.2f} |"
    for k, r in sets.items():
        rd, hd = dev[k]
        ms = timed(bvh, rd, hd, len(r), a.steps)
        abi.check_errors(0)
        got = abi.from_device(hd, F.HIT1).tobytes()
        ref.setdefault(k, got)
        line += f" {k} {ms:.4f} ms {len(r) / ms / 1e3:6.0f} Mrays/s identical {got == ref[k]} |"
    print(line, flush=True)
    del bvh
