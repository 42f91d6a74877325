#!/usr/bin/env python
"""bench.py -- headline benchmark: Mrays/s of the HIP BVH traversal (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Step  = one closest-hit traversal pass over one 1 048 576-ray batch resident in HBM
        (config[1] of BASELINE.json: <scene>.bvh + <scene>-primary.rays, tmax 5000).
        The random batch (config[2], tmax 1) is timed the same way and reported in "extra".
Scene = "sponza" if data/sponza.{bvh,-primary.rays,-random.rays} were supplied, otherwise the
        regenerable procedural "atrium" (the reference checkout lacks the Sponza blobs).
value = rays traced by all ranks per second / 1e6, kernel passes only (rays, BVH and hit
        buffers resident in HBM; H2D/D2H excluded like bench_traversal.cpp:124-135).
N > 1 = weak scaling, no data-path collective: the BVH is replicated, rank r traces sub-pixel
        sample r of N through the same 1024x1024 pixel grid (primary) / seed 42 + r (random);
        hit counts are all-gathered over RCCL after the timed region as a cross-check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # README.md:34-37 uses --warmup 10 --bench 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default=None)
    ap.add_argument("--bvh-width", type=int, default=int(os.environ.get("RODENT_BENCH_WIDTH", "2")), choices=(2, 8))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("RODENT_BENCH_VARIANT", "-1")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", choices=("primary", "random"), default=None, help="profiling aid: time only one ray set")
    return ap.parse_args()


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/rNN_traffic.json, written by
    scripts/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs; FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None if no profile of this kernel has been recorded."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_traffic.json")):
        try:
            data = json.loads(f.read_text())
        except ValueError:
            continue
        for name, t in data.items():
            if name.replace(" ", "").startswith(kernel.replace(" ", "").rstrip(">")) and "hbm_bytes_fetch_x2" in t:
                best = int(t["hbm_bytes_fetch_x2"])
    return best


def measured_issue(kernel, kernel_ms):
    """VALU occupancy of `kernel` on the primary pass from the committed SQ counter pass (profiles/rNN_traffic.json):
    fraction of the launch during which the 1024 SIMDs issue VALU instructions (4 cycles per wave64 instruction at
    2.4 GHz) and the fraction of lanes active in them.  None if no profile of this kernel has been recorded."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_traffic.json")):
        try:
            data = json.loads(f.read_text())
        except ValueError:
            continue
        for name, t in data.items():
            if name.replace(" ", "").startswith(kernel.replace(" ", "").rstrip(">")) and "SQ_ACTIVE_INST_VALU" in t and "SQ_THREAD_CYCLES_VALU" in t:
                best = {"valu_instructions_per_launch": int(t.get("SQ_INSTS_VALU", 0)),
                        "valu_busy_frac": round(4.0 * t["SQ_ACTIVE_INST_VALU"] / (kernel_ms * 1e-3 * 2.4e9 * 1024), 4),
                        "lane_utilisation": round(t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_ACTIVE_INST_VALU"]), 4),
                        "source": f.name}
    return best


def time_passes(abi, torch, bvh, rays_dev, hits_dev, n, variant, steps, warmup, dist, any_hit=False):
    """W untimed + K timed launches.  Returns (wall seconds for K steps [max over ranks is taken by
    the caller], mean kernel ms from HIP events recorded on the launch stream)."""
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        starts[i].record(stream)
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
        ends[i].record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    return wall, float(np.mean(kernel_ms)), float(np.median(kernel_ms)), float(np.min(kernel_ms))


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher the contract describes (one rank per GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29517"), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    import torch
    from rodent_amd import abi, formats as F, raygen, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        dist = dist_mod
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP traversal has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = local_rank

    # ---- inputs (rank 0 builds the files, the others wait) ----------------------------------
    scene = args.scene or scenes.default_scene()
    if rank == 0:
        bvh_path = scenes.scene_bvh(scene)
    if dist is not None:
        dist.barrier()
    bvh_path = scenes.scene_bvh(scene)
    width = args.bvh_width
    variant = args.variant if args.variant >= 0 else int(os.environ.get(f"RODENT_HIP_BVH{width}_VARIANT", "0"))
    bvh = abi.DeviceBvh.load(bvh_path, width, dev)

    if scene == "sponza":
        prim = F.read_rays(scenes.DATA / "sponza-primary.rays", 0.0, scenes.PRIMARY_TMAX)
        rnd = F.read_rays(scenes.DATA / "sponza-random.rays", 0.0, scenes.RANDOM_TMAX)
    else:
        eye, d, up, fov = scenes.CAMERAS[scene]
        prim = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX, sample=rank, num_samples=world)
        n4, _ = F.read_bvh(bvh_path, F.BVH4_TRI4)
        lo, hi = raygen.scene_bounds(n4)
        rnd = raygen.random_rays(lo, hi, 1 << 20, 42 + rank, 0.0, scenes.RANDOM_TMAX)
    n = len(prim)

    prim_dev, rnd_dev = abi.to_device(prim, dev), abi.to_device(rnd, dev)
    hits_dev = torch.zeros(n * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
    hits_rnd_dev = torch.zeros(len(rnd) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")

    # ---- timed region ------------------------------------------------------------------------
    steps_p, warm_p = (args.steps, args.warmup) if args.only != "random" else (1, 0)
    steps_r, warm_r = (args.steps, args.warmup) if args.only != "primary" else (1, 0)
    wall, k_mean, k_med, k_min = time_passes(abi, torch, bvh, prim_dev, hits_dev, n, variant, steps_p, warm_p, dist)
    wall_r, kr_mean, kr_med, kr_min = time_passes(abi, torch, bvh, rnd_dev, hits_rnd_dev, len(rnd), variant, steps_r, warm_r, dist)
    # for information only (never `value`): independent batches in flight on two streams -- the fill of one launch
    # overlaps the drain of the other (every (device, stream) has its own launch state)
    overlapped = None
    if args.only is None and not args.no_cpu_baseline:           # not in the profiling runs: their per-kernel averages are the serial launches
        try:
            s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
            h2 = [hits_dev, torch.zeros_like(hits_dev)]
            for k in range(2):
                abi.traverse_async(bvh, prim_dev, h2[k], n, False, variant, s2[k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, args.steps // 2)):
                for k in range(2):
                    abi.traverse_async(bvh, prim_dev, h2[k], n, False, variant, s2[k])
            torch.cuda.synchronize()
            overlapped = 2 * max(1, args.steps // 2) * n / (time.perf_counter() - t0) / 1e6
        except Exception as e:                                    # informational only: never lose the bench line over it
            print(f"bench.py: two-stream measurement skipped ({e})", file=sys.stderr)
    abi.lib()  # keep the handle alive
    if dist is not None:
        t = torch.tensor([wall, wall_r], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, wall_r = float(t[0]), float(t[1])
    value = n * world * steps_p / wall / 1e6
    value_rnd = len(rnd) * world * steps_r / wall_r / 1e6

    # ---- cross-check after the timed region: one gather of the per-rank hit counts -------------
    hits = abi.from_device(hits_dev, F.HIT1)
    hits_rnd = abi.from_device(hits_rnd_dev, F.HIT1)
    counts = torch.tensor([int((hits["tri_id"] >= 0).sum()), int((hits_rnd["tri_id"] >= 0).sum())], device=f"cuda:{dev}")
    if dist is not None:
        gathered = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts)
        counts_all = [g.tolist() for g in gathered]
    else:
        counts_all = [counts.tolist()]

    if rank != 0:
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- rank 0: algorithmic bytes (oracle visit counts), roofline, CPU baseline ---------------
    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * wall / steps_p, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene}.bvh + {scene}-primary.rays (1024x1024 primary rays, tmax 5000, closest hit) per GPU",
                   "rays_per_gpu_per_step": n, "bvh_layout": f"BVH{width}", "kernel": abi.kernel_name(width, variant),
                   "variant": abi.variants(width)[variant], "parallelism": f"replicated BVH x {world}, rays sharded by sub-pixel sample"},
        "extra": {"random_Mrays_s": round(value_rnd, 3), "random_ms_per_step": round(1e3 * wall_r / steps_r, 5),
                  "primary_kernel_ms": {"mean": round(k_mean, 5), "median": round(k_med, 5), "min": round(k_min, 5)},
                  "random_kernel_ms": {"mean": round(kr_mean, 5), "median": round(kr_med, 5), "min": round(kr_min, 5)},
                  "hit_counts_per_rank[primary,random]": counts_all,
                  "two_streams_Mrays_s_per_gpu": None if overlapped is None else round(overlapped, 3)},
    }
    if not args.no_cpu_baseline:
        from oracle import binding as O      # checker / CPU baseline only: never on the measured path
        # (1) visit counts of the reference algorithm for THIS layout -> algorithmic bytes per ray
        if width == 2:
            nodes, tris = F.read_bvh(bvh_path, F.BVH2_TRI1)
            node_b, prim_b, algo = 64, 48, "ref"
        else:
            nodes, tris = F.read_bvh(bvh_path, F.BVH8_TRI4)
            node_b, prim_b, algo = 256, 224, "gpu"
        sample = slice(0, n, 4)                                  # every 4th ray: 262 144 rays, deterministic
        ref_hits, st = O.traverse(width, nodes, tris, prim[sample], algo=algo)
        bytes_per_ray = 32 + 16 + node_b * st["inner_per_ray"] + prim_b * st["prims_per_ray"]
        achieved = bytes_per_ray * n / (k_mean * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": measured_traffic(abi.kernel_name(width, variant)),
                           "bytes_per_ray": round(bytes_per_ray, 2),
                           "visits_per_ray": {"inner": round(st["inner_per_ray"], 3), "prim": round(st["prims_per_ray"], 3)},
                           "compulsory_bytes_per_ray": 48, "kernel_ms": round(k_mean, 5),
                           "issue": measured_issue(abi.kernel_name(width, variant), k_mean)}
        # parity spot check on the sample (bit-exact for the order-preserving kernels)
        same = hits[sample].tobytes() == ref_hits.tobytes()
        out["extra"]["sample_bit_exact_vs_oracle"] = bool(same)
        if not same:
            ids_equal = float((hits[sample]["tri_id"] == ref_hits["tri_id"]).mean())
            out["extra"]["sample_id_match_fraction"] = ids_equal
    if world == 1 and not args.no_cpu_baseline:
        # (2) CPU baseline: Rodent's CPU hybrid path (ray8 x bvh8 packets with single-ray fallback,
        #     mapping_cpu.impala:259-402) restated with AVX2 (oracle/hybrid_baseline.cpp), timed on this host:
        #     once on 1 core (the reference's bench loop is sequential) and once on all hardware threads.
        n8, t8 = F.read_bvh(bvh_path, F.BVH8_TRI4)
        O.cpu_baseline(n8, t8, prim[:4096])                        # build / warm up
        threads = max(1, O.hardware_threads())
        t0 = time.perf_counter(); cpu_hits = O.cpu_baseline(n8, t8, prim, mode="hybrid", threads=1); cpu1_s = time.perf_counter() - t0
        t0 = time.perf_counter(); O.cpu_baseline(n8, t8, prim, mode="hybrid", threads=threads); cpuN_s = time.perf_counter() - t0
        t0 = time.perf_counter(); O.cpu_baseline(n8, t8, rnd, mode="hybrid", threads=threads); cpuN_rnd_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / cpuN_s / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"all {n} primary rays, 1 pass, hybrid ray8 x BVH8/Tri4 restatement of "
                                         "mapping_cpu.impala:259-402 (AVX2+FMA, -O3), dynamic 2048-ray chunks over all hardware threads"}
        out["extra"]["cpu_baseline_1core_Mrays_s"] = round(n / cpu1_s / 1e6, 3)
        out["extra"]["cpu_baseline_random_Mrays_s"] = round(len(rnd) / cpuN_rnd_s / 1e6, 3)
        out["extra"]["cpu_vs_gpu_hit_mismatch"] = int(((cpu_hits["tri_id"] >= 0) != (hits[:len(cpu_hits)]["tri_id"] >= 0)).sum())
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
