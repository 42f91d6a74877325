#!/usr/bin/env python
"""bench.py -- headline benchmark: Mrays/s of the HIP BVH traversal (BASELINE.json), renderer frame rates beside it.

  python bench.py --gpus N --steps K --warmup W [--weak]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Step  = one closest-hit traversal pass over one 1 048 576-ray batch resident in HBM (config[1] of BASELINE.json: <scene>.bvh +
        <scene>-primary.rays, tmax 5000).  The random batch (config[2], tmax 1) is timed the same way and reported in "extra".
Scene = "sponza" if data/sponza.{bvh,-primary.rays,-random.rays} were supplied, otherwise the regenerable procedural "atrium" (the
        reference checkout lacks the Sponza blobs).
value = rays traced by all ranks per second / 1e6, kernel passes only (rays, BVH and hit buffers resident in HBM; H2D / D2H excluded
        like bench_traversal.cpp:124-135).
N > 1 = no data-path collective: the BVH is replicated, rays are independent units.  BOTH partitions are timed:
        strong (`value`, "scaling": "strong" -- BASELINE's metric is ONE 1 Mi-ray dump at 1 / 2 / 4 / 8 GPUs): rank r
        traces the contiguous range ray_range(n, r, N) of the SAME set the N = 1 run traces; after the timed region one RCCL gather brings
        the Hit1 ranges to rank 0, which compares the assembled array with its own trace of the whole set (`extra.strong_scaling_check`);
        weak (`extra.weak_scaling`, `config.weak_scaling_Mrays_s`; `--weak` makes it `value`): rank r traces sub-pixel sample r of N
        through the same 1024 x 1024 pixel grid (primary) / seed 42 + r (random): 1 Mi rays per GPU per step, per-GPU work fixed.
        ONE 1 Mi-ray launch is latency-bound (its longest rays do not shard): `config.predicted_scaling_x` holds what one GPU predicts.
roofline = ONE bound, stated once: VALU issue.  achieved = VALU wave-instructions per launch (SQ_INSTS_VALU of the committed
        counter pass of THIS kernel on THESE sources) / live kernel time / SIMDs; peak = the guide's 2 cycles per wave64 VALU instruction
        at the clock measured in the calibration loop (MI355X_MICROARCH.md: 1 162 wave-instructions per us per SIMD at 2 323 MHz); frac =
        achieved / peak.  Beside it, each one division away from a file under profiles/: `frac_of_measured_loop_mix_ceiling` (the same
        rate against the microbenchmarked ceiling of the loop's own instruction classes), `lane_utilisation`, `roofline.hbm` (BASELINE's
        "fraction of HBM roofline": measured_frac = `traffic`, the FETCH_SIZE x 2 + WRITE_SIZE bytes of separate --pmc passes, / kernel
        time / 8 TB/s; compulsory_frac, traffic_over_compulsory, write_amplification) and `cache_served_bytes_over_hbm_peak` (SURVEY
        8(d)'s bytes per ray x rays / kernel time / 8 TB/s: > 1, because the 22 MB BVH is served by LDS / L1 / L2 / MALL -- a count of
        cache hits, no fraction, labelled so).  Counter-derived figures are only quoted while the profile's source hash matches the
        kernels' sources (rodent_amd/provenance.py); a stale profile is reported as such and the live node-fetch bound stands in.
render = `extra.render`: BASELINE configs 4 and 5 through the renderer ABI -- Cornell 1920 x 1080, 64 spp, path length 4 and the config-5
        scene at 3840 x 2160, 256 spp, path length 8 (one GPU: the whole frame; N GPUs: interleaved 16-row tiles + one film gather to
        rank 0), streaming and megakernel mappings, Msamples/s = spp * w * h / frame seconds / 1e6 (driver.cpp:300).
Code  = this file holds the contract (arguments, ranks, the JSON line); benchlib/ holds the sections: timing.py (the timed region),
        traversal.py (partitions, side measurements, CPU baseline), render.py (renderer section), profiles.py (profiles, rooflines).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# noqa: E402,F401
from benchlib.profiles import (HBM_PEAK_GBPS, binding_bounds, current_profile, kernel_counters, latest_json, measured_traffic,
                               pick_bound, render_profile, traversal_roofline)
# noqa: E402,F401  (scripts/ import these from here)
from benchlib.render import RENDER_CONFIGS, render_cpu_baseline, render_section, scene_file
from benchlib.traversal import Bench, cpu_baseline, scene_matrix_rows, side_measurements, timed_partitions  # noqa: E402

# what one GPU predicts for N (every rank's share timed alone: profiles/r06_range_costs.txt, r05_range_costs.txt, r04_band_costs.txt): ONE 1
# Mi-ray set does not shard its tail -- with the cost balanced every rank still holds a chunk of ~190 wave iterations
PREDICTED_SCALING = {"strong_1Mi_primary_contiguous_ranges": {"2": 1.27, "4": 1.63, "8": 1.90},
    "strong_1Mi_random": {"2": 1.33, "4": 1.72, "8": 2.05},
                     "cfg5_frame_interleaved_16_row_tiles": {"2": 1.99, "4": 3.99, "8": 7.94}, "weak": "1 Mi rays per GPU: ~N",
                     "source": "profiles/r06_range_costs.txt, profiles/r05_range_costs.txt, profiles/r04_band_costs.txt"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # README.md:34-37 uses --warmup 10 --bench 50
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default=None)
    ap.add_argument("--bvh-width", type=int, default=int(os.environ.get("RODENT_BENCH_WIDTH", "2")), choices=(2, 4, 8))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("RODENT_BENCH_VARIANT", "-1")))
    ap.add_argument("--weak", action="store_true", help="N > 1: report the weak-scaling figure (1 Mi rays per GPU) as `value` instead")
    ap.add_argument("--strong", action="store_true",
                    help="accepted for compatibility: strong scaling (ONE 1 Mi-ray set in contiguous ranges, SURVEY 8e) is the default for "
                        "N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true",
        help="profiling aid: no oracle leg, no CPU baselines, no informational extras")
    ap.add_argument("--no-render", action="store_true", help="skip the renderer section (extra.render)")
    ap.add_argument("--no-scenes", action="store_true",
                    help="skip the scene x ray-class matrix (extra.scenes: the gallery / crown / plant classes are generated and built "
                        "first, ~80 s)")
    ap.add_argument("--render-spp5", type=int, default=256, help="samples per pixel of the config-5 frame (BASELINE: 256)")
    ap.add_argument("--only", choices=("primary", "random"), default=None, help="profiling aid: time only one ray set")
    return ap.parse_args()


def init_ranks(args):
    """(torch, torch.distributed or None, rank, local rank = device, world size).  Started as plain `python bench.py --gpus N`: becomes the
    launcher the contract describes (one rank per GPU)."""
    if args.gpus > 1 and "RANK" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29517"), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    import torch
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE",
        "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RODENT_BENCH_BACKEND=gloo RODENT_BENCH_SHARE_GPUS=1 (test mode: the N-rank code path on a box with fewer GPUs than ranks -- ranks
        # share devices, collectives go through the host); the driver's runs use RCCL, one rank per GPU
        backend = os.environ.get("RODENT_BENCH_BACKEND", "nccl")
        if os.environ.get("RODENT_BENCH_SHARE_GPUS", "0") not in ("", "0"):
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP traversal has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    return torch, dist, rank, local_rank, world


def main():
    args = parse_args()
    torch, dist, rank, dev, world = init_ranks(args)
    from rodent_amd import abi, formats as F, scenes

    # ---- inputs (rank 0 builds the files, the others wait) ----------------------------------
    scene = args.scene or scenes.default_scene()
    if rank == 0:
        scenes.scene_bvh(scene)
    if dist is not None:
        dist.barrier()
    bvh_path = scenes.scene_bvh(scene)
    width = args.bvh_width
    variant = args.variant if args.variant >= 0 else int(os.environ.get(f"RODENT_HIP_BVH{width}_VARIANT", "0"))
    b = Bench(args=args, torch=torch, abi=abi, dist=dist, rank=rank, world=world, dev=dev, scene=scene, bvh_path=bvh_path, width=width,
              variant=variant, bvh=abi.DeviceBvh.load(bvh_path, width, dev), info=not args.no_cpu_baseline and args.only is None,
              steps={"primary": (args.steps, args.warmup) if args.only != "random" else (1, 0),
                     "random": (args.steps, args.warmup) if args.only != "primary" else (1, 0)})
    steps_p, steps_r = b.steps["primary"][0], b.steps["random"][0]

    # ---- timed regions, then what rides along ------------------------------------------------
    part, scaling, scaling_extra = timed_partitions(b)
    prim, rnd, n = part["prim"], part["rnd"], len(part["prim"])
    (k_mean, k_med, k_min), (kr_mean, kr_med, kr_min) = part["k"], part["kr"]
    side = side_measurements(b, part)
    hits = abi.from_device(part["hits_dev"], F.HIT1)[:n]
    hits_rnd = abi.from_device(part["hits_rnd_dev"], F.HIT1)[:len(rnd)]
    scene_rows = scene_matrix_rows(b)
    # renderer (BASELINE configs 4 / 5): after the traversal's timed regions, own timed frames
    render = render_section(args, torch, dist, rank, world, dev) if b.info and not args.no_render and width == 2 else None
    if rank != 0:
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- rank 0: the JSON line -----------------------------------------------------------------
    kname = abi.kernel_name(width, variant)
    sharding = {"strong": "ONE ray set in contiguous ranges (strong scaling), Hit1 gather to rank 0 after the timed region",
                "weak": "rays sharded by sub-pixel sample (weak scaling)"}[scaling]
    backend = None if dist is None else ("nccl (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend() + " (test mode)")
    out = {
        "metric": "Mrays/s", "value": round(part["value"], 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup,
        "ms_per_step": round(1e3 * part["wall"] / steps_p, 5), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{scene}.bvh + {scene}-primary.rays (1024x1024 primary rays, tmax 5000, closest hit)"
                               + (" per GPU" if scaling == "weak" and world > 1 else ""),
                   "rays_per_gpu_per_step": n, "bvh_layout": f"BVH{width}", "kernel": kname, "variant": abi.variants(width)[variant],
                   "parallelism": f"replicated BVH x {world}" + (f", {sharding}" if world > 1 else ""), "world_size": world,
                   "collective_backend": backend},
        "extra": {"random_Mrays_s": round(part["value_rnd"], 3), "random_ms_per_step": round(1e3 * part["wall_r"] / steps_r, 5),
                  # mean: HIP events around the whole timed region / steps; single_*: each launch between its own event pair in a second
                  # pass (adds the dispatch latency)
                  "primary_kernel_ms": {"mean": round(k_mean, 5), "single_launch_median": round(k_med, 5),
                      "single_launch_min": round(k_min, 5)},
                  "random_kernel_ms": {"mean": round(kr_mean, 5), "single_launch_median": round(kr_med, 5),
                      "single_launch_min": round(kr_min, 5)},
                  "kernel_ms_per_rank[primary,random]": part["kernel_ms_per_rank"],
                  "hit_counts[primary,random]": [int((hits["tri_id"] >= 0).sum()), int((hits_rnd["tri_id"] >= 0).sum())],
                  # the prebuilt .so against the sources lying next to it
                  "library": {"version": abi.lib().rodent_hip_version().decode(),
                      "source_digest": abi.lib().rodent_hip_source_digest().decode(),
                              "built_from_these_sources": abi.built_from_these_sources()},
                  "two_streams_Mrays_s_per_gpu": None, "primary_16Mi_rays_per_launch": None, "random_8Mi_rays_per_launch": None},
    }
    cfg, extra = out["config"], out["extra"]
    extra.update(side)
    extra.update(scaling_extra)
    if render is not None:
        extra["render"] = render
    if scene_rows:
        extra["scenes"] = scene_rows
        extra["scenes_what"] = (
            "1 Mi camera rays / random segments (closest hit) / ao rays (ray_gen shadow, any hit, tmax 0.999) per scene class through "
                                "the default mapping (top) and 'fast' / 'refill' beside it: kernel ms, Mrays/s of the default, oracle "
                                    "parity of a 32 Ki-ray "
                                "sample, oracle visits per ray and stack depths, blocks spilled")
        classes = ("primary", "random", "ao")
        cfg["ao_Mrays_s"] = (scene_rows.get(scene, {}).get("ao") or {}).get("Mrays_s")
        cfg["scene_classes_Mrays_s[primary,random,ao]"] = {k: [(v.get(c) or {}).get("Mrays_s") for c in classes]
                                                           for k, v in scene_rows.items() if "primary" in v}
        cfg["scene_classes_parity"] = all(v[c]["sample_parity"] for v in scene_rows.values() if "primary" in v for c in classes)
    # what the driver's record keeps is `config`, `roofline` and `cpu_baseline`: the other headline figures as short scalars
    # stateless: the in-kernel choice alone (the ray-kind hint is off by default)
    cfg["random_Mrays_s"] = round(part["value_rnd"], 1)
    if "random_with_kind_hint" in side:
        cfg["random_with_kind_hint_Mrays_s"] = side["random_with_kind_hint"]["Mrays_s"]
    if "primary_in_list_order" in side:
        cfg["primary_in_list_order_Mrays_s"] = side["primary_in_list_order"]["Mrays_s"]
    if world > 1:
        cfg["predicted_scaling_x"] = PREDICTED_SCALING
        cfg["world_size_seen_by_the_collective_backend"] = dist.get_world_size()
        strong_rec, weak = scaling_extra["strong_scaling"], scaling_extra["weak_scaling"]
        # ONE 1 Mi-ray set over the N GPUs
        cfg["strong_scaling_Mrays_s[primary,random]"] = [strong_rec["Mrays_s"], strong_rec["random_Mrays_s"]]
        cfg["weak_scaling_Mrays_s[primary,random]"] = [weak["Mrays_s"], weak["random_Mrays_s"]]
    if side.get("primary_16Mi_rays_per_launch"):
        cfg["primary_16Mi_Mrays_s"] = side["primary_16Mi_rays_per_launch"].get("Mrays_s")
    if side.get("random_8Mi_rays_per_launch"):
        cfg["random_8Mi_Mrays_s"] = side["random_8Mi_rays_per_launch"].get("default_Mrays_s")
    for cfg_name, entry in (render or {}).items():
        auto = entry.get("auto") or {}
        if "Msamples_s" in auto:
            cfg[cfg_name.split("_")[0] + "_Msamples_s"] = auto["Msamples_s"]
            cfg[cfg_name.split("_")[0] + "_mapping"] = entry.get("auto_mapping")
    if not args.no_cpu_baseline:
        out["roofline"], extra["all_rays_bit_exact_vs_oracle"] = traversal_roofline(b, part, hits, hits_rnd, kname)
        cfg["hbm_measured_frac"] = out["roofline"]["hbm_measured_frac"]
        cfg["cache_served_bytes_over_hbm_peak"] = out["roofline"]["cache_served_bytes_over_hbm_peak"]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], more, threads = cpu_baseline(b, prim, rnd, hits)
        extra.update(more)
        if render is not None:
            try:
                extra["render"]["cpu_baseline"] = render_cpu_baseline(threads)
            except Exception as e:                                # informational: never lose the bench line over it
                extra["render"]["cpu_baseline"] = {"skipped": str(e)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
