"""Traversal section of bench.py: the ray sets, the timed partitions (one GPU; N GPUs: strong and weak), and the measurements that ride
along in `extra` / `config` (never `value`): mappings with state, the other mappings of BASELINE config 3, larger launches, two streams,
the scene x ray-class matrix."""
from __future__ import annotations

import sys
import time
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .timing import gather_scalars, max_over_ranks, time_passes

ROOT = Path(__file__).resolve().parent.parent


@dataclass
class Bench:
    """What every section needs: the arguments, the modules that need a GPU (imported once by bench.py), the ranks, the hierarchy."""
    args: object
    torch: object
    abi: object
    dist: object          # torch.distributed, or None with one rank
    rank: int
    world: int
    dev: int
    scene: str
    bvh_path: Path
    width: int
    variant: int
    bvh: object
    info: bool            # the informational extras run (never in the profiling runs)
    steps: dict = field(default_factory=dict)      # {"primary": (K, W), "random": (K, W)}

    @property
    def default_top(self):
        """The shipped BVH2 default mapping is what runs (its switches and side measurements apply)."""
        return self.width == 2 and self.abi.variants(2)[self.variant] == "top"


def ray_sets(b: Bench, sample, samples):
    """(camera rays, random segments): 1024 x 1024 pixels through sub-pixel sample `sample` of `samples`, 1 Mi segments with seed 42 +
    sample."""
    from rodent_amd import formats as F, raygen, scenes
    if b.scene == "sponza":
        return (F.read_rays(scenes.DATA / "sponza-primary.rays", 0.0, scenes.PRIMARY_TMAX),
                F.read_rays(scenes.DATA / "sponza-random.rays", 0.0, scenes.RANDOM_TMAX))
    eye, d, up, fov = scenes.CAMERAS[b.scene]
    p = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX, sample=sample, num_samples=samples)
    n4, _ = F.read_bvh(b.bvh_path, F.BVH4_TRI4)
    lo, hi = raygen.scene_bounds(n4)
    return p, raygen.random_rays(lo, hi, 1 << 20, 42 + sample, 0.0, scenes.RANDOM_TMAX)


def run_partition(b: Bench, prim, rnd):
    """Times both sets on this rank's share; returns the per-partition record (times already max over ranks)."""
    from rodent_amd import formats as F
    torch, abi, dev = b.torch, b.abi, b.dev
    (steps_p, warm_p), (steps_r, warm_r) = b.steps["primary"], b.steps["random"]
    pd, rd = abi.to_device(prim, dev), abi.to_device(rnd, dev)
    hp = torch.zeros(max(len(prim), 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
    hr = torch.zeros(max(len(rnd), 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
    wall, k_mean, k_med, k_min = time_passes(abi, torch, b.bvh, pd, hp, len(prim), b.variant, steps_p, warm_p, b.dist)
    wall_r, kr_mean, kr_med, kr_min = time_passes(abi, torch, b.bvh, rd, hr, len(rnd), b.variant, steps_r, warm_r, b.dist)
    abi.check_errors(dev)
    wall, wall_r = max_over_ranks(torch, b.dist, dev, [wall, wall_r])
    per_rank = gather_scalars(torch, b.dist, dev, [k_mean, kr_mean, float(len(prim)), float(len(rnd))])
    total_p, total_r = int(sum(g[2] for g in per_rank)), int(sum(g[3] for g in per_rank))
    return {"prim": prim, "rnd": rnd, "prim_dev": pd, "rnd_dev": rd, "hits_dev": hp, "hits_rnd_dev": hr, "wall": wall, "wall_r": wall_r,
            "k": (k_mean, k_med, k_min), "kr": (kr_mean, kr_med, kr_min), "total": total_p, "total_rnd": total_r,
            "value": total_p * steps_p / wall / 1e6, "value_rnd": total_r * steps_r / wall_r / 1e6,
            "kernel_ms_per_rank": [[round(g[0], 5), round(g[1], 5)] for g in per_rank]}


def timed_partitions(b: Bench):
    """The timed regions.  One rank: the whole 1 Mi-ray sets.  N ranks: the strong partition (ONE set in contiguous ranges -- `value` unless
    --weak) and the weak one (1 Mi rays per rank).  Returns (the partition `value` is quoted on, "strong" | "weak", extras for N > 1)."""
    from rodent_amd import formats as F, parallel
    abi, args = b.abi, b.args
    steps_p = b.steps["primary"][0]
    prim_all, rnd_all = ray_sets(b, 0, 1)
    if b.world == 1:
        # (N = 1 of the fixed 1 Mi-ray workload: the same set the N > 1 runs split)
        return run_partition(b, prim_all, rnd_all), "strong", {}
    # strong: ONE ray set in contiguous ranges -- contiguous keeps coherent rays coherent
    lo, hi = parallel.ray_range(len(prim_all), b.rank, b.world)
    strong = run_partition(b, prim_all[lo:hi], rnd_all[lo:hi])
    # after the timed region: ONE gather of the Hit1 ranges to rank 0 (16 B/ray: 16 MiB in total), compared there with a single-GPU
    # trace of the whole set
    full = parallel.gather_hits_device(strong["hits_dev"], len(prim_all), b.dist, b.dev)
    full_rnd = parallel.gather_hits_device(strong["hits_rnd_dev"], len(rnd_all), b.dist, b.dev)
    check = None
    if b.rank == 0:
        whole = abi.traverse(b.bvh, prim_all, variant=b.variant)
        whole_rnd = abi.traverse(b.bvh, rnd_all, variant=b.variant)
        check = {"primary_equal_to_single_gpu": bool(full.tobytes() == whole.tobytes()),
                 "random_equal_to_single_gpu": bool(full_rnd.tobytes() == whole_rnd.tobytes())}
    # weak: every rank its own 1 Mi rays (sub-pixel sample `rank` of `world`)
    pw, rw = ray_sets(b, b.rank, b.world)
    weak_part = run_partition(b, pw, rw)
    counts = gather_scalars(b.torch, b.dist, b.dev, [float((abi.from_device(weak_part["hits_dev"], F.HIT1)["tri_id"] >= 0).sum()),
                                                      float((abi.from_device(weak_part["hits_rnd_dev"], F.HIT1)["tri_id"] >= 0).sum())])
    weak = {"Mrays_s": round(weak_part["value"], 3), "ms_per_step": round(1e3 * weak_part["wall"] / steps_p, 5),
            "random_Mrays_s": round(weak_part["value_rnd"], 3), "rays_per_gpu_per_step": len(pw),
            "kernel_ms_per_rank[primary,random]": weak_part["kernel_ms_per_rank"],
            "hit_counts_per_rank[primary,random]": [[int(c[0]), int(c[1])] for c in counts],
            "what": "rank r traces sub-pixel sample r of N through the same 1024 x 1024 pixel grid (random: seed 42 + r): per-GPU work "
                "fixed"}
    strong_rec = {"Mrays_s": round(strong["value"], 3), "ms_per_step": round(1e3 * strong["wall"] / steps_p, 5),
                  "random_Mrays_s": round(strong["value_rnd"], 3), "rays_per_gpu_per_step": len(strong["prim"]),
                  "kernel_ms_per_rank[primary,random]": strong["kernel_ms_per_rank"],
                  "what": "ONE 1 Mi-ray set in contiguous ranges (SURVEY 8e), Hit1 gather to rank 0 after the timed region"}
    main_part, scaling = (weak_part, "weak") if args.weak and not args.strong else (strong, "strong")
    return main_part, scaling, {"strong_scaling": strong_rec, "weak_scaling": weak, "strong_scaling_check": check}


def side_measurements(b: Bench, part):
    """The same ray sets through what is NOT the headline: mappings that carry state between launches (schedule history, ray-kind hint),
    BASELINE config 3's other readings ("sorted", "refill"), the default without its tile mapping, two streams, 16 Mi / 8 Mi-ray launches.
    One rank only.  Returns {key in `extra`: record}."""
    from rodent_amd import formats as F, raygen, scenes
    torch, abi, args, bvh, variant, width, dev = b.torch, b.abi, b.args, b.bvh, b.variant, b.width, b.dev
    out = {}
    if b.world != 1:
        return out
    (steps_p, warm_p), (steps_r, warm_r) = b.steps["primary"], b.steps["random"]
    n, n_r = len(part["prim"]), len(part["rnd"])
    prim_dev, rnd_dev, hits_dev, hits_rnd_dev = part["prim_dev"], part["rnd_dev"], part["hits_dev"], part["hits_rnd_dev"]

    def timed(rays_dev, count, v, steps, warm):
        hits = torch.zeros(max(count, 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
        wall, k_mean, _, _ = time_passes(abi, torch, bvh, rays_dev, hits, count, v, steps, warm, None)
        return hits, round(count * steps / wall / 1e6, 3), round(1e3 * wall / steps, 5), round(k_mean, 5)

    # the schedule history (rodent_hip_schedule_history: chunks traced longest first by the previous launch's per-chunk cost -- state
    # carried from step to step, therefore NOT the headline; same hit records)
    if b.default_top and b.info:
        abi.lib().rodent_hip_schedule_history(1)
        hp, vp, msp, kp = timed(prim_dev, n, variant, steps_p, warm_p)
        hr, vr, msr, kr = timed(rnd_dev, n_r, variant, steps_r, warm_r)
        abi.lib().rodent_hip_schedule_history(0)
        torch.cuda.synchronize()
        out["with_schedule_history"] = {
            "primary_Mrays_s": vp, "primary_ms_per_step": msp, "primary_kernels_ms": kp, "random_Mrays_s": vr, "random_ms_per_step": msr,
            "random_kernels_ms": kr, "identical_hits": bool(torch.equal(hp, hits_dev) and torch.equal(hr, hits_rnd_dev)),
            "what": "rodent_hip_schedule_history(1): every launch records the wave iterations of each 64-ray chunk, the next launch of the "
                "same "
                    "size traces its chunks longest first; off by default, not the headline value"}
    # BASELINE config 3 ("ray compaction/sorting on"): the random set through "sorted" (the permutation by origin cell is rebuilt inside
    # every timed launch) and through "refill" (continuous compaction inside the persistent kernel)
    if args.only != "primary":
        for name, key, same_key, what in (
                ("sorted", "random_sorted", "identical_to_unsorted",
                 "sorted: counting sort of the rays on 512 Morton cells of their origin inside every launch, then the default kernel "
                     "through "
                 "the permutation"),
                ("refill", "random_refill", "identical_to_default",
                 "refill: the default's persistent workgroups; a wave whose idle lanes reach 32 draws that many new rays from its stripe's "
                 "counter")):
            if name in abi.variants(width):
                h, v, ms, k = timed(rnd_dev, n_r, abi.variants(width).index(name), steps_r, warm_r)
                out[key] = {"Mrays_s": v, "ms_per_step": ms, "kernels_ms": k, "variant": what, same_key: bool(torch.equal(h, hits_rnd_dev))}
        # the default WITH the ray-kind hint (rodent_hip_ray_kind_hint(1); off by default from round 5 on): state carried from launch to
        # launch
        if b.default_top:
            abi.ray_kind_hint(True)
            h, v, _, k = timed(rnd_dev, n_r, variant, steps_r, warm_r)
            abi.ray_kind_hint(False)
            out["random_with_kind_hint"] = {
                "Mrays_s": v, "kernels_ms": k, "identical_to_default": bool(torch.equal(h, hits_rnd_dev)),
                "what": "rodent_hip_ray_kind_hint(1): the list goes to k_bvh2_top_refill from its second launch on; the default "
                        "(random_Mrays_s) is k_bvh2_top_auto alone, whose waves find their rays incoherent and run the refill loop -- no "
                            "state "
                        "between launches"}
    # the default WITHOUT the tile mapping (rodent_hip_ray_grid(0): camera rays in list order, 64 pixels of a row per wavefront, as until
    # round 4).  Not in the profiling runs: the same kernel name in another mode would mix into their per-kernel means.
    if b.default_top and args.only != "random" and not args.no_cpu_baseline:
        abi.ray_grid(0)
        h, v, _, k = timed(prim_dev, n, variant, steps_p, warm_p)
        abi.ray_grid(-1)
        out["primary_in_list_order"] = {
            "Mrays_s": v, "kernels_ms": k, "identical_to_default": bool(torch.equal(h, hits_dev)),
            "what": "rodent_hip_ray_grid(0): the same kernel with the wave's 64 rays in list order (64 pixels of an image row); the "
                    "default recognises the image width from 66 of the launch's rays and gives every wavefront an 8 x 8-pixel tile -- no "
                    "state between launches, hit records identical"}
    abi.check_errors(dev)                                         # the asynchronous entry points report stack overflows through a flag
    if not b.info:
        return out
    # for information only: independent batches in flight on two streams -- the fill of one launch overlaps the drain of the other
    # (every (device, stream) has its own launch state)
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
        out["two_streams_Mrays_s_per_gpu"] = round(2 * max(1, args.steps // 2) * n / (time.perf_counter() - t0) / 1e6, 3)
    except Exception as e:                                        # informational only: never lose the bench line over it
        print(f"bench.py: two-stream measurement skipped ({e})", file=sys.stderr)
    if b.scene == "sponza":
        return out
    # the same camera at 4096 x 4096 = 16 Mi primary rays per launch -- the kernel's throughput regime (at 1 Mi rays the launch is bound
    # by its schedule, LAB_NOTES.md 3.1.1)
    try:
        eye, d, up, fov = scenes.CAMERAS[b.scene]
        big_rays = raygen.primary_rays(eye, d, up, fov, 4096, 4096, 0.0, scenes.PRIMARY_TMAX)
        big_dev = abi.to_device(big_rays, dev)
        _, v, ms, k = timed(big_dev, len(big_rays), variant, 10, 3)
        out["primary_16Mi_rays_per_launch"] = {"rays_per_launch": len(big_rays), "Mrays_s": v, "ms_per_step": ms, "kernels_ms": k}
        del big_dev, big_rays
    except Exception as e:
        print(f"bench.py: 16 Mi-ray measurement skipped ({e})", file=sys.stderr)
    # ... and 8 Mi random segments per launch (what a renderer's bounce pass looks like): the default mapping and "refill"
    try:
        if width == 2 and "refill" in abi.variants(width):
            lo8, hi8 = raygen.scene_bounds(F.read_bvh(b.bvh_path, F.BVH4_TRI4)[0])
            rnd8 = raygen.random_rays(lo8, hi8, 1 << 23, 43, 0.0, scenes.RANDOM_TMAX)
            rnd8_dev = abi.to_device(rnd8, dev)
            h_d, v_d, _, k_d = timed(rnd8_dev, len(rnd8), variant, 10, 3)
            h_f, v_f, _, k_f = timed(rnd8_dev, len(rnd8), abi.variants(width).index("refill"), 10, 3)
            out["random_8Mi_rays_per_launch"] = {"rays_per_launch": len(rnd8), "default_Mrays_s": v_d, "default_kernels_ms": k_d,
                                                 "refill_Mrays_s": v_f, "refill_kernels_ms": k_f,
                                                     "identical_hits": bool(torch.equal(h_d, h_f))}
            del rnd8_dev, h_d, h_f, rnd8
    except Exception as e:
        print(f"bench.py: 8 Mi random-ray measurement skipped ({e})", file=sys.stderr)
    return out


def scene_matrix_rows(b: Bench):
    """The other scene classes and the any-hit ray class (scripts/scene_matrix.py; the oracle checks a 32 Ki-ray sample
    of every cell).  None when it does not apply."""
    if not (b.info and b.world == 1 and b.width == 2 and not b.args.no_scenes and b.scene != "sponza"):
        return None
    rows = None
    try:
        sys.path.insert(0, str(ROOT / "scripts"))
        import scene_matrix                                          # lab tooling; imports the oracle as its checker
        rows, t0 = {}, time.time()
        for name in (b.scene, "gallery", "crown", "plant"):
            if time.time() - t0 > 300:                               # a slow host must not cost the bench line
                rows[name] = {"skipped": "the scenes before this one took more than 300 s to build and trace"}
                continue
            rows[name] = scene_matrix.measure(name, steps=b.args.steps, quiet=True)
    except Exception as e:                                          # informational: never lose the bench line over it
        print(f"bench.py: scene matrix skipped ({e})", file=sys.stderr)
        rows = rows or None
    return rows


def cpu_baseline(b: Bench, prim, rnd, hits):
    """Rodent's CPU hybrid path (ray8 x bvh8 packets with single-ray fallback, mapping_cpu.impala:259-402) restated with AVX2
    (oracle/hybrid_baseline.cpp), timed on this host on a PERSISTENT thread pool (threads created outside the timed region), median of
    the passes.  Returns (the `cpu_baseline` object, extras, threads used)."""
    from oracle import binding as O      # checker / CPU baseline only: never on the measured path
    from rodent_amd import formats as F
    n = len(prim)
    n8, t8 = F.read_bvh(b.bvh_path, F.BVH8_TRI4)
    threads = max(1, O.hardware_threads())
    passes = 12
    secs, cpu_hits = O.cpu_baseline_bench(n8, t8, prim, threads, passes)
    secs_rnd, _ = O.cpu_baseline_bench(n8, t8, rnd, threads, passes)
    secs1, _ = O.cpu_baseline_bench(n8, t8, prim, 1, 2)
    record = {"value": round(n / float(np.median(secs)) / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
              "sample": f"all {n} primary rays x {passes} passes (median; one warm-up pass before), hybrid ray8 x BVH8/Tri4 restatement of "
                        "mapping_cpu.impala:259-402 (AVX2+FMA, -O3), persistent pool of all hardware threads pulling 1024-ray chunks",
              "passes": passes, "pass_ms": [round(1e3 * float(s), 3) for s in secs]}
    extra = {"cpu_baseline_1core_Mrays_s": round(n / float(np.median(secs1)) / 1e6, 3),
             "cpu_baseline_random_Mrays_s": round(len(rnd) / float(np.median(secs_rnd)) / 1e6, 3),
             "cpu_vs_gpu_hit_mismatch": int(((cpu_hits["tri_id"] >= 0) != (hits[:len(cpu_hits)]["tri_id"] >= 0)).sum())}
    return record, extra, threads
