#!/usr/bin/env python
"""CPU model of wave-packet traversal for coherent 64-ray chunks (scripts/model_packet.c does the counting).

For every 64-ray chunk of a ray dump the C model traverses the BVH2 as ONE packet with a shared stack (node fetched once per
wave, every lane tests both child boxes, ballots decide; reference: src/traversal/mapping_cpu.impala:259-384) and hands subtrees
that fewer than T lanes enter to the existing per-lane single-step loop -- at once (mode "immediate", the reference's order,
:305-321) or after the packet phase, all lanes together (mode "deferred").  T = 65 IS the existing kernel; its hits must equal
oracle B1 bit for bit (checked), and its modelled launch time is what the price list below is calibrated against.

Printed per (mode, T): packet steps and per-lane wave iterations per chunk, active lanes per packet visit, modelled VALU
wave-instructions per launch (the throughput proxy: the 16 Mi-ray regime), the longest chunk's dependent chain, and a modelled
launch time at the dump's size from a fluid schedule (1024 SIMDs x 8 wave slots; a wave alone issues one VALU instruction every
5.7 cycles and waits for its loads, a full SIMD issues one every 3.1 cycles -- profiles/r02_ubench_valu_peak.txt), chunks drawn
in the default kernel's ticket order.  Parity against B1: rays whose t / tri_id differ (packet order changes which box is culled
by an already-shortened tmax; exact ties resolve differently).

usage: python scripts/model_packet.py [--rays data/atrium-primary.rays --tmax 5000] [--tile 8x8 --width 1024] [--any]
"""
import argparse
import ctypes as C
import heapq
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import binding as O
from rodent_amd import formats as F, topimage

ap = argparse.ArgumentParser()
ap.add_argument("--bvh", default=str(ROOT / "data/atrium.bvh"))
ap.add_argument("--rays", default=str(ROOT / "data/atrium-primary.rays"))
ap.add_argument("--tmin", type=float, default=0.0)
ap.add_argument("--tmax", type=float, default=5000.0)
ap.add_argument("--any", action="store_true")
ap.add_argument("--tile", default="", help="THxTW: re-chunk the dump (a WIDTH-wide image in scanline order) into pixel tiles first")
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--thresholds", default="8,16,24,32,48")
ap.add_argument("--limit", type=int, default=0, help="only the first N rays")
ap.add_argument("--buildable", action="store_true",
    help="also mode 2: no masks on the shared stack, fallback decided at the parent, at most --keep-limit kept entries per lane, 15-entry "
    "windows")
ap.add_argument("--keep-limit", type=int, default=8)
ap.add_argument("--steal", default="",
    help="i0:every[,i0:every...]: also mode 3, the per-lane kernel with work stealing inside the wave from iteration i0 on, every `every` "
    "iterations")
a = ap.parse_args()

so = Path("/tmp/model_packet.so")
src = ROOT / "scripts/model_packet.c"
if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, (ROOT / "oracle/traversal_oracle.c").stat().st_mtime):
    subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-shared", str(src), "-lm", "-o",
        str(so)], check=True)
lib = C.CDLL(str(so))
COUNTS = np.dtype([(k, "<u4") for k in ("p_node_img", "p_node_mem", "p_tri", "p_lanes_node", "p_lanes_tri", "f_it_node", "f_it_mixed",
    "f_it_tri",
                                        "f_lane_steps", "f_phases", "max_deferred", "window_overflows")])
lib.model_packet.restype = C.c_int
lib.model_packet.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p] * 3

nodes, tris = F.read_bvh(a.bvh, F.BVH2_TRI1)
rays = F.read_rays(a.rays, a.tmin, a.tmax)
if a.limit:
    rays = rays[:a.limit]
if a.tile:
    th, tw = (int(x) for x in a.tile.split("x"))
    w = a.width; h = len(rays) // w
    rows, cols = h // th, w // tw
    t = np.arange(rows * cols); tr, tc = t // cols, t % cols
    jy, jx = np.arange(th * tw) // tw, np.arange(th * tw) % tw
    perm = (((tr[:, None] * th + jy[None, :]) * w) + tc[:, None] * tw + jx[None, :]).ravel()
    rays = np.ascontiguousarray(rays[perm])
n = len(rays)
chunks = (n + 63) // 64
in_image = np.zeros(len(nodes), np.uint8)
in_image[topimage.image_nodes(nodes)] = 1
ref, stats = O.traverse(2, nodes, tris, rays, any_hit=a.any)
print(f"{a.rays}{' tiles ' + a.tile if a.tile else ''}: {n} rays, {chunks} chunks; B1 visits per ray: {stats['inner_per_ray']:.2f} inner "
    f"nodes, {stats['prims_per_ray']:.2f} triangles")

# ---- price list (VALU wave-instructions; cycles of a wave running alone) --------------------------------------------------
# per-lane loop: scripts/isa_hist.py on k_bvh2_top_persist: 67 node lanes only, 132 both kinds; triangle lanes only ~ 132 -
# 67 + 10
V_LANE = {"node": 67, "mixed": 132, "tri": 75}
# packet steps (estimates from the operation list: 6 v_pk_fma + 12 min/max + 2 cmp for the two slabs, the near-child preference,
# mask bookkeeping on the SALU; a triangle: 3 sub, 2 cross, 4 dot, prodsign, compares, one IEEE division, 3 mul, 4 cndmask)
V_PNODE, V_PTRI = 30, 55
CYC_VALU_ALONE, CYC_VALU_FULL = 5.7, 3.1
# load latency a step waits for when the wave runs alone (cycles at 2.4 GHz): the per-lane iteration's 1250 cycles = 100 VALU x 5.7 + ~680
LAT_LANE = 680
# ds_read broadcast + SALU decisions; scalar / uniform load (children prefetched); triangle record
LAT_PNODE_IMG, LAT_PNODE_MEM, LAT_PTRI = 230, 450, 300
GHZ = 2.4


def price(c):
    c = {k: c[k].astype(np.float64) for k in c.dtype.names}
    it = c["f_it_node"] + c["f_it_mixed"] + c["f_it_tri"]
    valu = V_LANE["node"] * c["f_it_node"] + V_LANE["mixed"] * c["f_it_mixed"] + V_LANE["tri"] * c["f_it_tri"] \
        + V_PNODE * (c["p_node_img"] + c["p_node_mem"]) + V_PTRI * c["p_tri"]
    lat = LAT_LANE * it + LAT_PNODE_IMG * c["p_node_img"] + LAT_PNODE_MEM * c["p_node_mem"] + LAT_PTRI * c["p_tri"]
    return valu, valu * CYC_VALU_ALONE + lat                       # VALU instructions, cycles alone


def ticket_order(total_chunks):
    """chunk order of k_bvh2_top_persist: ticket t of stripe s = chunk ((t / 32) * 64 + s) * 32 + t % 32, stripes round-robin"""
    order = []
    t = 0
    while len(order) < total_chunks:
        for s in range(64):
            ch = ((t // 32) * 64 + s) * 32 + t % 32
            if ch < total_chunks:
                order.append(ch)
        t += 1
    return order


def schedule(valu, alone, simds=1024, slots=8):
    """fluid model: a SIMD's waves progress at their solo speed while their summed issue demand fits the SIMD, else scaled down"""
    order = ticket_order(len(valu))
    nxt = 0
    now = [0.0] * simds
    waves = [[] for _ in range(simds)]                       # per SIMD: [remaining solo cycles, demand]
    # initial fill: chunk k of the order goes to SIMD k % simds (workgroups spread over the chip)
    for k in range(min(len(order), simds * slots)):
        ch = order[k]
        waves[k % simds].append([alone[ch], valu[ch] * CYC_VALU_FULL / max(alone[ch], 1.0)])
    nxt = min(len(order), simds * slots)

    def next_event(s):
        ws = waves[s]
        if not ws:
            return None
        speed = 1.0 / max(1.0, sum(w[1] for w in ws))
        return now[s] + min(w[0] for w in ws) / speed

    heap = [(next_event(s), s) for s in range(simds) if waves[s]]
    heapq.heapify(heap)
    end = 0.0
    while heap:
        t_ev, s = heapq.heappop(heap)
        ws = waves[s]
        speed = 1.0 / max(1.0, sum(w[1] for w in ws))
        adv = (t_ev - now[s]) * speed
        now[s] = t_ev
        for w in ws:
            w[0] -= adv
        keep = [w for w in ws if w[0] > 1e-6]
        freed = len(ws) - len(keep)
        for _ in range(freed):
            if nxt < len(order):
                ch = order[nxt]; nxt += 1
                keep.append([alone[ch], valu[ch] * CYC_VALU_FULL / max(alone[ch], 1.0)])
        waves[s] = keep
        end = max(end, t_ev)
        ev = next_event(s)
        if ev is not None:
            heapq.heappush(heap, (ev, s))
    return end / (GHZ * 1e3)                                  # microseconds


def run(mode, T):
    hits = np.zeros(n, F.HIT1)
    counts = np.zeros(chunks, COUNTS)
    hist = np.zeros(65, np.uint64)
    rc = lib.model_packet(nodes.ctypes.data, tris.ctypes.data, rays.ctypes.data, hits.ctypes.data, n, int(a.any), mode, T,
                          in_image.ctypes.data, counts.ctypes.data, hist.ctypes.data)
    assert rc == 0, "model stack overflow"
    return hits, counts, hist


profile = np.zeros(1024, np.uint64)
lib.model_set_iteration_profile.argtypes = [C.c_void_p]
lib.model_set_iteration_profile(profile.ctypes.data)
base_hits, base_counts, _ = run(1, 65)
lib.model_set_iteration_profile(None)
assert base_hits.tobytes() == ref.tobytes(), "T = 65 must reproduce oracle B1"
bv, ba = price(base_counts)
bt = schedule(bv, ba)
bit = base_counts["f_it_node"].astype(np.int64) + base_counts["f_it_mixed"] + base_counts["f_it_tri"]
print(f"existing kernel (T = 65; hits == B1: yes): wave iterations per chunk {bit.mean():.1f} (longest {bit.max()}), lane utilisation "
    f"{base_counts['f_lane_steps'].sum() / (64.0 * bit.sum()):.3f}, "
      f"VALU {bv.sum() / 1e6:.1f} M wave-instructions, longest chunk alone {ba.max() / GHZ / 1e3:.1f} us, modelled launch {bt:.1f} us")
kinds = {k: int(base_counts["f_it_" + k].sum()) for k in ("node", "mixed", "tri")}
print("  its iterations by kind: " + ", ".join(f"{k} {v / bit.sum():.1%}" for k, v in kinds.items()))
lanes_by_it, its_by_it = profile[:512].astype(np.float64), profile[512:].astype(np.float64)
print("  where its lanes idle -- iteration index: share of the launch's wave iterations, lanes active in them")
for lo, hi in ((0, 10), (10, 20), (20, 30), (30, 40), (40, 60), (60, 100), (100, 512)):
    w = its_by_it[lo:hi].sum()
    print(f"    iterations {lo:3d}..{hi - 1:3d}: {w / its_by_it.sum():6.1%} of the wave iterations at "
        f"{lanes_by_it[lo:hi].sum() / max(1.0, 64.0 * w):5.1%} of the lanes")
print()
hdr = f"{'mode':10s} {'T':>3s} | {'pkt node img/mem':>17s} {'pkt tri':>8s} {'lanes/visit':>11s} | {'lane iters':>10s} {'(longest)':>9s} " \
    f"{'lane util':>9s} {'max kept':>8s} | {'VALU M':>8s} {'vs now':>6s} | {'longest alone us':>16s} {'launch us':>9s} {'vs now':>6s} | " \
    f"{'t differs':>9s} {'id differs':>10s}"
print(hdr)
# the shared tmax is read from LDS every iteration; a stealing event moves 13 ray registers through ds_bpermute
V_STEAL_PER_ITERATION, V_STEAL_EVENT = 3, 45
ap_modes = [(("immediate", 0, int(x)), ("deferred", 1, int(x))) for x in a.thresholds.split(",") if x]
ap_modes = [m for pair in zip(*ap_modes) for m in pair] if ap_modes else []
ap_modes = sorted(ap_modes, key=lambda m: m[1])
if a.buildable:
    ap_modes += [("buildable", 2, int(x)) for x in a.thresholds.split(",") if x]
for spec in a.steal.split(","):
    if spec:
        i0, every = (int(x) for x in spec.split(":"))
        ap_modes.append((f"steal {i0}/{every}", 3, i0 + (every << 8)))
for mode_name, mode, T in ap_modes:
    if True:
        hits, c, hist = run(mode, T + (a.keep_limit << 8 if mode == 2 else 0))
        v, al = price(c)
        if mode == 3:
            extra = V_STEAL_PER_ITERATION * (c["f_it_node"].astype(np.float64) + c["f_it_mixed"] + c["f_it_tri"]) + V_STEAL_EVENT * c[
                "f_phases"]
            v, al = v + extra, al + extra * CYC_VALU_ALONE
        tl = schedule(v, al)
        it = c["f_it_node"].astype(np.int64) + c["f_it_mixed"] + c["f_it_tri"]
        visits = c["p_node_img"].astype(np.int64) + c["p_node_mem"] + c["p_tri"]
        lanes_per_visit = (c["p_lanes_node"].sum() + c["p_lanes_tri"].sum()) / max(1, visits.sum())
        if a.any:
            tdiff = int(((hits["tri_id"] >= 0) != (ref["tri_id"] >= 0)).sum()); iddiff = 0
        else:
            tdiff = int((hits["t"].view(np.uint32) != ref["t"].view(np.uint32)).sum())
            iddiff = int((hits["tri_id"] != ref["tri_id"]).sum())
        print(f"{mode_name:10s} {T & 255:3d} | {c['p_node_img'].mean():8.1f}/{c['p_node_mem'].mean():8.1f} {c['p_tri'].mean():8.1f} "
            f"{lanes_per_visit:11.1f} | {it.mean():10.1f} {it.max():9d} "
              f"{c['f_lane_steps'].sum() / max(1.0, 64.0 * it.sum()):9.3f} "
                  f"{str(c['max_deferred'].max()) + ('/' + str(int(c['window_overflows'].sum())) if mode == 2 else ''):>8s} | "
                  f"{v.sum() / 1e6:8.1f} {bv.sum() / v.sum():6.2f} | {al.max() / GHZ / 1e3:16.1f} {tl:9.1f} {bt / tl:6.2f} | {tdiff:9d} "
                  f"{iddiff:10d}")
