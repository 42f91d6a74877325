#!/usr/bin/env python
"""bench.py -- headline benchmark: Mrays/s of the HIP BVH traversal (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Step  = one closest-hit traversal pass over one 1 048 576-ray batch resident in HBM
        (config[1] of BASELINE.json: <scene>.bvh + <scene>-primary.rays, tmax 5000).
        The random batch (config[2], tmax 1) is timed the same way and reported in "extra".
Scene = "sponza" if data/sponza.{bvh,-primary.rays,-random.rays} were supplied, otherwise the
        regenerable procedural "atrium" (the reference checkout lacks the Sponza blobs).
value = rays traced by all ranks per second / 1e6, kernel passes only (rays, BVH and hit
        buffers resident in HBM; H2D/D2H excluded like bench_traversal.cpp:124-135).
N > 1 = no data-path collective: the BVH is replicated.
        default (weak scaling): rank r traces sub-pixel sample r of N through the same 1024x1024 pixel grid (primary) /
        seed 42 + r (random): 1 Mi rays per GPU per step;
        --strong: ONE 1 Mi-ray set, rank r traces the contiguous range ray_range(n, r, N); after the timed
        region one RCCL all-gather collects the Hit1 ranges and rank 0 compares the assembled array with its own trace
        of the whole set.
roofline: "bound: hbm" is SURVEY 8(d)'s algorithmic-bytes figure (it exceeds 1: the 22 MB BVH is served by L1 / L2 / MALL,
        not by HBM).  The bounds that DO bind this kernel are reported next to it: the node-fetch rate of
        the vector-memory pipeline (TA -> L1 -> L2) and the VALU issue rate, both against peaks measured on this chip by
        the microbenchmarks under scripts/ubench (profiles/rNN_calibration.json), plus measured HBM traffic.
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
LDS_CLOCK_GHZ = 2.4             # same guide: 2.4 GHz engine clock


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # README.md:34-37 uses --warmup 10 --bench 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default=None)
    ap.add_argument("--bvh-width", type=int, default=int(os.environ.get("RODENT_BENCH_WIDTH", "2")), choices=(2, 4, 8))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("RODENT_BENCH_VARIANT", "-1")))
    ap.add_argument("--strong", action="store_true", help="N > 1: shard ONE ray set in contiguous ranges (strong scaling) and gather the Hit1 array")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", choices=("primary", "random"), default=None, help="profiling aid: time only one ray set")
    return ap.parse_args()


def latest_json(pattern):
    best = None
    for f in sorted((ROOT / "profiles").glob(pattern)):
        try:
            best = (f.name, json.loads(f.read_text()))
        except ValueError:
            continue
    return best


def kernel_counters(kernel, ray_set):
    """Per-launch counter means of `kernel` on the `ray_set` pass from the committed PMC passes (profiles/rNN_pmc_counters.json,
    scripts/profile_pmc.sh: one small counter group per rocprofv3 --pmc run).  {} if this kernel was not profiled."""
    found = latest_json("r*_pmc_counters.json")
    out = {}
    if found:
        name, data = found
        key = kernel.replace(" ", "").rstrip(">")
        for group, kernels in data.items():
            if f"_{ray_set}_" not in group:
                continue
            for k, counters in kernels.items():
                if k.replace(" ", "").startswith(key):
                    out.update(counters)
        if out:
            out["source"] = name
    return out


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (primary pass): FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes
    (profiles/rNN_traffic.json, scripts/profile_round.sh); FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950."""
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


def binding_bounds(kernel, ray_set, rays, steps_per_ray, kernel_ms, lds_steps_per_ray=0.0):
    """The bounds that bind the traversal kernel, against peaks MEASURED on this chip (profiles/rNN_calibration.json):
    node / triangle fetches per ns through the vector-memory pipeline (live: oracle visit counts x rays / HIP-event kernel
    time; `lds_steps_per_ray` of them are served from the LDS image of the default mapping instead and are reported against
    the LDS rate) and VALU wave-instructions per us per SIMD (instruction count from the committed SQ counter pass / live
    kernel time)."""
    cal = latest_json("r*_calibration.json")
    if not cal:
        return None
    cal_name, cal = cal
    c = kernel_counters(kernel, ray_set)
    fetches_per_ns = (steps_per_ray - lds_steps_per_ray) * rays / (kernel_ms * 1e6)
    if ray_set == "primary":
        peak, peak_kind = cal["node_fetch_peak_coherent"], "64-byte node per lane, neighbouring lanes share nodes, L1/L2-resident (vmem_peak 'coherent')"
    else:
        # incoherent rays: every lane its own node; blend of the scattered-L2 and scattered-MALL rates by the measured L2 hit rate
        hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else 0.85
        peak = 1.0 / (hit / cal["node_fetch_peak_scattered_l2"] + (1.0 - hit) / cal["node_fetch_peak_scattered_mall"])
        peak_kind = f"64-byte node per lane, scattered: L2 rate x {hit:.3f} + MALL rate x {1 - hit:.3f} (measured TCC hit rate; vmem_peak 'scattered')"
    out = {"node_fetch": {"bound": "vector-memory pipeline (TA/L1/L2), node fetches", "unit": "fetches/ns", "achieved": round(fetches_per_ns, 2), "peak": round(peak, 2),
                          "frac": round(fetches_per_ns / peak, 4), "peak_kind": peak_kind, "steps_per_ray": round(steps_per_ray - lds_steps_per_ray, 3), "peak_source": cal_name}}
    if lds_steps_per_ray:
        # MI355X_MICROARCH.md, LDS table: ds_read_b128 4 and ds_read_b64 2 LDS cycles per wave-instruction when conflict-free -> 3 x 4 + 2 per 64 node records
        lds_peak = LDS_CLOCK_GHZ * cal["cus"] * 64 / 14.0
        lds_per_ns = lds_steps_per_ray * rays / (kernel_ms * 1e6)
        out["lds_fetch"] = {"bound": "LDS (top-of-tree image: 3 x ds_read_b128 + ds_read_b64 per node)", "unit": "fetches/ns", "achieved": round(lds_per_ns, 2),
                            "peak": round(lds_peak, 1), "frac": round(lds_per_ns / lds_peak, 4), "steps_per_ray": round(lds_steps_per_ray, 3),
                            "peak_kind": "conflict-free rate of the guide's LDS table at 2.4 GHz; distinct records on one bank quarter serialise"}
    if "SQ_INSTS_VALU" in c:
        per_simd_us = c["SQ_INSTS_VALU"] / (kernel_ms * 1e3) / cal["simds"]
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        out["valu_issue"] = {"bound": "VALU issue", "unit": "wave-instructions/us/SIMD", "achieved": round(per_simd_us, 1), "peak": cal["valu_issue_peak"],
                             "frac": round(per_simd_us / cal["valu_issue_peak"], 4), "valu_instructions_per_launch": int(c["SQ_INSTS_VALU"]),
                             "lane_utilisation": round(lane_util, 4), "useful_lane_frac": round(per_simd_us / cal["valu_issue_peak"] * lane_util, 4),
                             "peak_source": cal_name, "counter_source": c.get("source")}
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
        out["ta_busy_frac_profiled"] = round(c["TA_TA_BUSY_sum"] / cal["cus"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)     # mean over the 256 TAs / cycles of one XCD
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out["wave_cycles_waiting_frac_profiled"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    return out


def time_passes(abi, torch, bvh, rays_dev, hits_dev, n, variant, steps, warmup, dist, any_hit=False):
    """W untimed + K timed launches.  Returns (wall seconds for K steps [max over ranks is taken by
    the caller], mean / median / min kernel ms from HIP events recorded on the launch stream)."""
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
    from rodent_amd import abi, formats as F, parallel, raygen, scenes

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
    strong = args.strong and world > 1

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

    sample, samples = (0, 1) if strong else (rank, world)
    if scene == "sponza":
        prim_all = F.read_rays(scenes.DATA / "sponza-primary.rays", 0.0, scenes.PRIMARY_TMAX)
        rnd_all = F.read_rays(scenes.DATA / "sponza-random.rays", 0.0, scenes.RANDOM_TMAX)
    else:
        eye, d, up, fov = scenes.CAMERAS[scene]
        prim_all = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX, sample=sample, num_samples=samples)
        n4, _ = F.read_bvh(bvh_path, F.BVH4_TRI4)
        lo, hi = raygen.scene_bounds(n4)
        rnd_all = raygen.random_rays(lo, hi, 1 << 20, 42 + sample, 0.0, scenes.RANDOM_TMAX)
    if strong:                                                   # contiguous ranges keep coherent rays coherent
        a, b = parallel.ray_range(len(prim_all), rank, world)
        prim, rnd = prim_all[a:b], rnd_all[a:b]
    else:
        prim, rnd = prim_all, rnd_all
    n = len(prim)

    prim_dev, rnd_dev = abi.to_device(prim, dev), abi.to_device(rnd, dev)
    hits_dev = torch.zeros(n * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
    hits_rnd_dev = torch.zeros(len(rnd) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")

    # ---- timed region ------------------------------------------------------------------------
    steps_p, warm_p = (args.steps, args.warmup) if args.only != "random" else (1, 0)
    steps_r, warm_r = (args.steps, args.warmup) if args.only != "primary" else (1, 0)
    wall, k_mean, k_med, k_min = time_passes(abi, torch, bvh, prim_dev, hits_dev, n, variant, steps_p, warm_p, dist)
    wall_r, kr_mean, kr_med, kr_min = time_passes(abi, torch, bvh, rnd_dev, hits_rnd_dev, len(rnd), variant, steps_r, warm_r, dist)
    # the same two sets with the schedule history on (rodent_hip_schedule_history: chunks traced longest first by the previous
    # launch's per-chunk cost -- state carried from step to step, therefore NOT the headline; same hit records)
    history = None
    if width == 2 and abi.variants(2)[variant] == "top" and world == 1 and args.only is None:
        abi.lib().rodent_hip_schedule_history(1)
        hits_h, hits_rnd_h = torch.zeros_like(hits_dev), torch.zeros_like(hits_rnd_dev)
        wall_h, kh_mean, _, _ = time_passes(abi, torch, bvh, prim_dev, hits_h, n, variant, steps_p, warm_p, None)
        wall_hr, khr_mean, _, _ = time_passes(abi, torch, bvh, rnd_dev, hits_rnd_h, len(rnd), variant, steps_r, warm_r, None)
        abi.lib().rodent_hip_schedule_history(0)
        torch.cuda.synchronize()
        history = {"primary_Mrays_s": round(n * steps_p / wall_h / 1e6, 3), "primary_ms_per_step": round(1e3 * wall_h / steps_p, 5), "primary_kernels_ms": round(kh_mean, 5),
                   "random_Mrays_s": round(len(rnd) * steps_r / wall_hr / 1e6, 3), "random_ms_per_step": round(1e3 * wall_hr / steps_r, 5), "random_kernels_ms": round(khr_mean, 5),
                   "identical_hits": bool(torch.equal(hits_h, hits_dev) and torch.equal(hits_rnd_h, hits_rnd_dev)),
                   "what": "rodent_hip_schedule_history(1): every launch records the wave iterations of each 64-ray chunk, the next launch of the same size traces its "
                           "chunks longest first; off by default, not the headline value"}
    # BASELINE config 3 ("ray compaction/sorting on"): the same random set through the "sorted" mapping -- the permutation by
    # origin cell is rebuilt inside every timed launch
    sorted_variant = abi.variants(width).index("sorted") if "sorted" in abi.variants(width) else None
    wall_s = ks_mean = None
    hits_sorted_dev = None
    if sorted_variant is not None and args.only != "primary":
        hits_sorted_dev = torch.zeros_like(hits_rnd_dev)
        wall_s, ks_mean, _, _ = time_passes(abi, torch, bvh, rnd_dev, hits_sorted_dev, len(rnd), sorted_variant, steps_r, warm_r, dist)
    abi.check_errors(dev)                                         # the asynchronous entry points report stack overflows through a flag
    # for information only (never `value`): independent batches in flight on two streams -- the fill of one launch
    # overlaps the drain of the other (every (device, stream) has its own launch state)
    overlapped = None
    if args.only is None and not args.no_cpu_baseline and world == 1:      # not in the profiling runs: their per-kernel averages are the serial launches
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
    # informational: the same camera at 4096 x 4096 = 16 Mi primary rays per launch -- the kernel's throughput regime (at 1 Mi rays
    # the launch is bound by its schedule)
    big = None
    if args.only is None and not args.no_cpu_baseline and world == 1 and scene != "sponza":
        try:
            eye, d, up, fov = scenes.CAMERAS[scene]
            big_rays = raygen.primary_rays(eye, d, up, fov, 4096, 4096, 0.0, scenes.PRIMARY_TMAX)
            big_dev = abi.to_device(big_rays, dev)
            big_hits = torch.zeros(len(big_rays) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
            wall_b, kb_mean, _, _ = time_passes(abi, torch, bvh, big_dev, big_hits, len(big_rays), variant, 10, 3, None)
            big = {"rays_per_launch": len(big_rays), "Mrays_s": round(len(big_rays) * 10 / wall_b / 1e6, 3), "ms_per_step": round(1e3 * wall_b / 10, 5), "kernels_ms": round(kb_mean, 5)}
            del big_dev, big_hits, big_rays
        except Exception as e:
            print(f"bench.py: 16 Mi-ray measurement skipped ({e})", file=sys.stderr)
    abi.lib()  # keep the handle alive
    kernel_ms_ranks = [[k_mean, kr_mean]]
    total_rays, total_rnd = n, len(rnd)
    if dist is not None:
        t = torch.tensor([wall, wall_r], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, wall_r = float(t[0]), float(t[1])
        km = torch.tensor([k_mean, kr_mean, float(n), float(len(rnd))], dtype=torch.float64, device=f"cuda:{dev}")
        gathered_km = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(gathered_km, km)
        kernel_ms_ranks = [[round(float(g[0]), 5), round(float(g[1]), 5)] for g in gathered_km]
        total_rays, total_rnd = int(sum(float(g[2]) for g in gathered_km)), int(sum(float(g[3]) for g in gathered_km))
    value = total_rays * steps_p / wall / 1e6
    value_rnd = total_rnd * steps_r / wall_r / 1e6
    if wall_s is not None and dist is not None:
        t = torch.tensor([wall_s], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_s = float(t[0])

    # ---- after the timed region: ONE gather --------------------------------------------------
    hits = abi.from_device(hits_dev, F.HIT1)
    hits_rnd = abi.from_device(hits_rnd_dev, F.HIT1)
    strong_check = None
    if strong:
        # the Hit1 ranges of all ranks, one RCCL all-gather (16 B/ray: 16 MiB in total), assembled in ray order on every rank
        full = parallel.gather_hits_device(hits_dev, len(prim_all), dist, dev)
        full_rnd = parallel.gather_hits_device(hits_rnd_dev, len(rnd_all), dist, dev)
        if rank == 0:
            whole = abi.traverse(bvh, prim_all, variant=variant)
            whole_rnd = abi.traverse(bvh, rnd_all, variant=variant)
            strong_check = {"primary_equal_to_single_gpu": bool(full.tobytes() == whole.tobytes()), "random_equal_to_single_gpu": bool(full_rnd.tobytes() == whole_rnd.tobytes())}
        counts_all = None
    else:
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

    # ---- rank 0: algorithmic bytes (oracle visit counts), rooflines, CPU baseline ---------------
    kname = abi.kernel_name(width, variant)
    sharding = "ONE ray set in contiguous ranges (strong scaling), Hit1 all-gather after the timed region" if strong else "rays sharded by sub-pixel sample (weak scaling)"
    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * wall / steps_p, 5), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene}.bvh + {scene}-primary.rays (1024x1024 primary rays, tmax 5000, closest hit)" + ("" if strong else " per GPU"),
                   "rays_per_gpu_per_step": n, "bvh_layout": f"BVH{width}", "kernel": kname,
                   "variant": abi.variants(width)[variant], "parallelism": f"replicated BVH x {world}, {sharding}",
                   "world_size": world, "collective_backend": "nccl (RCCL)" if world > 1 else None},
        "extra": {"random_Mrays_s": round(value_rnd, 3), "random_ms_per_step": round(1e3 * wall_r / steps_r, 5),
                  "primary_kernel_ms": {"mean": round(k_mean, 5), "median": round(k_med, 5), "min": round(k_min, 5)},
                  "random_kernel_ms": {"mean": round(kr_mean, 5), "median": round(kr_med, 5), "min": round(kr_min, 5)},
                  "kernel_ms_per_rank[primary,random]": kernel_ms_ranks,
                  "hit_counts_per_rank[primary,random]": counts_all, "strong_scaling_check": strong_check,
                  "two_streams_Mrays_s_per_gpu": None if overlapped is None else round(overlapped, 3),
                  "primary_16Mi_rays_per_launch": big},
    }
    if history is not None:
        out["extra"]["with_schedule_history"] = history
    if wall_s is not None:
        out["extra"]["random_sorted"] = {"Mrays_s": round(total_rnd * steps_r / wall_s / 1e6, 3), "ms_per_step": round(1e3 * wall_s / steps_r, 5), "kernels_ms": round(ks_mean, 5),
                                         "variant": "sorted: counting sort of the rays on 512 Morton cells of their origin inside every launch, then the default kernel through the permutation",
                                         "identical_to_unsorted": bool(abi.from_device(hits_sorted_dev, F.HIT1).tobytes() == hits_rnd.tobytes())}
    if not args.no_cpu_baseline:
        from oracle import binding as O      # checker / CPU baseline only: never on the measured path
        # (1) visit counts of the reference algorithm for THIS layout over ALL rays -> algorithmic bytes per ray; full parity check
        block = {2: F.BVH2_TRI1, 4: F.BVH4_TRI4, 8: F.BVH8_TRI4}[width]
        nodes, tris = F.read_bvh(bvh_path, block)
        node_b, prim_b, algo = {2: (64, 48, "ref"), 4: (128, 224, "gpu"), 8: (256, 224, "gpu")}[width]
        ref_hits, st = O.traverse(width, nodes, tris, prim, algo=algo)
        ref_rnd, st_r = O.traverse(width, nodes, tris, rnd, algo=algo)
        lds_p = lds_r = 0.0
        if width == 2 and abi.variants(2)[variant] == "top" and n >= 9216 * 64:
            # the share of the node visits that the default mapping serves from its LDS image (host restatement of the image's node set)
            from rodent_amd import topimage
            ids = topimage.image_nodes(nodes)
            lds_p = float(O.node_visits(nodes, tris, prim)[ids].sum()) / len(prim)
            lds_r = float(O.node_visits(nodes, tris, rnd)[ids].sum()) / len(rnd)
        bytes_per_ray = 32 + 16 + node_b * st["inner_per_ray"] + prim_b * st["prims_per_ray"]
        achieved = bytes_per_ray * n / (k_mean * 1e-3) / 1e9
        traffic = measured_traffic(kname)
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                           "note": "frac > 1: SURVEY 8(d)'s algorithmic bytes count every node / triangle visit, and the BVH is served by L1 / L2 / MALL -- HBM does not bind this kernel; see hbm_measured and binding",
                           "bytes_per_ray": round(bytes_per_ray, 2),
                           "visits_per_ray": {"inner": round(st["inner_per_ray"], 3), "prim": round(st["prims_per_ray"], 3)},
                           "compulsory_bytes_per_ray": 48, "kernel_ms": round(k_mean, 5),
                           "hbm_measured": None if traffic is None else {"bytes_per_launch": traffic, "GBps": round(traffic / (k_mean * 1e-3) / 1e9, 1), "frac": round(traffic / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                           "binding": binding_bounds(kname, "primary", n, st["inner_per_ray"] + st["prims_per_ray"], k_mean, lds_p),
                           "random": {"kernel_ms": round(kr_mean, 5), "visits_per_ray": {"inner": round(st_r["inner_per_ray"], 3), "prim": round(st_r["prims_per_ray"], 3)},
                                      "binding": binding_bounds(kname, "random", len(rnd), st_r["inner_per_ray"] + st_r["prims_per_ray"], kr_mean, lds_r)}}
        # parity on every ray of both sets (bit-exact for the order-preserving kernels)
        out["extra"]["all_rays_bit_exact_vs_oracle"] = {"primary": bool(hits.tobytes() == ref_hits.tobytes()), "random": bool(hits_rnd.tobytes() == ref_rnd.tobytes())}
    if world == 1 and not args.no_cpu_baseline:
        # (2) CPU baseline: Rodent's CPU hybrid path (ray8 x bvh8 packets with single-ray fallback,
        #     mapping_cpu.impala:259-402) restated with AVX2 (oracle/hybrid_baseline.cpp), timed on this host on a
        #     PERSISTENT thread pool (threads created outside the timed region), median of the passes.
        n8, t8 = F.read_bvh(bvh_path, F.BVH8_TRI4)
        threads = max(1, O.hardware_threads())
        passes = 12
        secs, cpu_hits = O.cpu_baseline_bench(n8, t8, prim, threads, passes)
        secs_rnd, _ = O.cpu_baseline_bench(n8, t8, rnd, threads, passes)
        secs1, _ = O.cpu_baseline_bench(n8, t8, prim, 1, 2)
        out["cpu_baseline"] = {"value": round(n / float(np.median(secs)) / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"all {n} primary rays x {passes} passes (median; one warm-up pass before), hybrid ray8 x BVH8/Tri4 restatement of "
                                         "mapping_cpu.impala:259-402 (AVX2+FMA, -O3), persistent pool of all hardware threads pulling 1024-ray chunks",
                               "passes": passes, "pass_ms": [round(1e3 * float(s), 3) for s in secs]}
        out["extra"]["cpu_baseline_1core_Mrays_s"] = round(n / float(np.median(secs1)) / 1e6, 3)
        out["extra"]["cpu_baseline_random_Mrays_s"] = round(len(rnd) / float(np.median(secs_rnd)) / 1e6, 3)
        out["extra"]["cpu_vs_gpu_hit_mismatch"] = int(((cpu_hits["tri_id"] >= 0) != (hits[:len(cpu_hits)]["tri_id"] >= 0)).sum())
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
