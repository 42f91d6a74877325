#!/usr/bin/env python
"""bench.py -- headline benchmark: Mrays/s of the HIP BVH traversal (BASELINE.json), renderer frame rates beside it.

  python bench.py --gpus N --steps K --warmup W [--strong]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Step  = one closest-hit traversal pass over one 1 048 576-ray batch resident in HBM
        (config[1] of BASELINE.json: <scene>.bvh + <scene>-primary.rays, tmax 5000).
        The random batch (config[2], tmax 1) is timed the same way and reported in "extra".
Scene = "sponza" if data/sponza.{bvh,-primary.rays,-random.rays} were supplied, otherwise the
        regenerable procedural "atrium" (the reference checkout lacks the Sponza blobs).
value = rays traced by all ranks per second / 1e6, kernel passes only (rays, BVH and hit
        buffers resident in HBM; H2D/D2H excluded like bench_traversal.cpp:124-135).
N > 1 = no data-path collective: the BVH is replicated, rays are independent units.  BOTH partitions are timed:
        strong (`value`, "scaling": "strong" -- BASELINE's metric is ONE 1 Mi-ray dump at 1 / 2 / 4 / 8 GPUs): rank r traces the
        contiguous range ray_range(n, r, N) of the SAME set the N = 1 run traces; after the timed region one RCCL gather brings the Hit1 ranges
        to rank 0, which compares the assembled array with its own trace of the whole set (`extra.strong_scaling_check`);
        weak (`extra.weak_scaling`, `config.weak_scaling_Mrays_s`; `--weak` makes it `value`): rank r traces sub-pixel sample r of N through the
        same 1024 x 1024 pixel grid (primary) / seed 42 + r (random): 1 Mi rays per GPU per step, per-GPU work fixed.
        ONE 1 Mi-ray launch is latency-bound (its longest rays do not shard): `config.predicted_scaling_x` holds what one GPU predicts.
roofline: ONE bound, stated once: VALU issue.  achieved = VALU wave-instructions per launch (SQ_INSTS_VALU of the committed counter
        pass of THIS kernel on THESE sources) / live kernel time / SIMDs; peak = the guide's 2 cycles per wave64 VALU instruction at the clock measured
        in the calibration loop (MI355X_MICROARCH.md: 1 162 wave-instructions per us per SIMD at 2 323 MHz); frac = achieved / peak.  Beside it, each
        one division away from a file under profiles/: `frac_of_measured_loop_mix_ceiling` (the same rate against the microbenchmarked ceiling of the
        loop's own instruction classes), `lane_utilisation`, `roofline.hbm` (BASELINE's "fraction of HBM roofline": `traffic`, the FETCH_SIZE x 2 +
        WRITE_SIZE bytes of separate --pmc passes, / kernel time / 8 TB/s = measured_frac; compulsory_frac, traffic_over_compulsory, write_amplification) and
        `cache_served_bytes_over_hbm_peak` (SURVEY 8(d)'s bytes per ray x rays / kernel time / 8 TB/s: > 1, because the 22 MB BVH is served by LDS / L1 / L2 /
        MALL -- a count of cache hits, no fraction, labelled so).  Counter-derived figures are only quoted while the profile's source
        hash matches the kernels' sources (rodent_amd/provenance.py); a stale profile is reported as such and the live node-fetch bound stands in.
render  (`extra.render`): BASELINE configs 4 and 5 through the renderer ABI -- Cornell 1920 x 1080, 64 spp, path length 4 and the
        config-5 scene at 3840 x 2160, 256 spp, path length 8 (one GPU: the whole frame; N GPUs: row bands + one film gather
        to rank 0), streaming and megakernel mappings, Msamples/s = spp * w * h / frame seconds / 1e6 (driver.cpp:300).
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
    ap.add_argument("--weak", action="store_true", help="N > 1: report the weak-scaling figure (1 Mi rays per GPU) as `value` instead")
    ap.add_argument("--strong", action="store_true", help="accepted for compatibility: strong scaling (ONE 1 Mi-ray set in contiguous ranges, SURVEY 8e) is the default for N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="profiling aid: no oracle leg, no CPU baselines, no informational extras")
    ap.add_argument("--no-render", action="store_true", help="skip the renderer section (extra.render)")
    ap.add_argument("--no-scenes", action="store_true", help="skip the scene x ray-class matrix (extra.scenes: the gallery / crown / plant classes are generated and built first, ~80 s)")
    ap.add_argument("--render-spp5", type=int, default=256, help="samples per pixel of the config-5 frame (BASELINE: 256)")
    ap.add_argument("--only", choices=("primary", "random"), default=None, help="profiling aid: time only one ray set")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# committed profiles (profiles/rNN_*.json) -- quoted only while they belong to the sources that are running
# ------------------------------------------------------------------------------------------------
def latest_json(pattern):
    best = None
    for f in sorted((ROOT / "profiles").glob(pattern)):
        try:
            best = (f.name, json.loads(f.read_text()))
        except ValueError:
            continue
    return best


def current_profile(pattern, kind):
    """(file name, data, None) of the newest committed profile matching `pattern` if it was taken on the present sources of
    `kind`, else (name or None, None, reason)."""
    from rodent_amd import provenance
    found = latest_json(pattern)
    if not found:
        return None, None, f"no profiles/{pattern}"
    name, data = found
    if not provenance.is_current(data.get("_meta"), kind):
        return name, None, f"profiles/{name} was taken on other {kind} sources (source_sha {data.get('_meta', {}).get('source_sha')} != {provenance.source_sha(kind)}): not quoted"
    return name, data, None


def kernel_counters(kernel, ray_set):
    """Per-launch counter means of `kernel` on the `ray_set` pass from the committed PMC passes (profiles/rNN_pmc_counters.json,
    scripts/profile_pmc.sh: one small counter group per rocprofv3 --pmc run).  ({}, reason) if there is no profile of this
    kernel on the present sources."""
    name, data, why = current_profile("r*_pmc_counters.json", "traversal")
    if data is None:
        return {}, why
    out = {}
    key = kernel.replace(" ", "").rstrip(">")
    for group, kernels in data.items():
        if group == "_meta" or f"_{ray_set}_" not in group:
            continue
        for k, counters in kernels.items():
            if k.replace(" ", "").startswith(key):
                out.update(counters)
    if not out:
        return {}, f"profiles/{name} holds no pass of kernel {kernel}"
    out["source"] = name
    return out, None


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (primary pass): FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes
    (profiles/rNN_traffic.json, scripts/profile_round.sh); FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950."""
    name, data, why = current_profile("r*_traffic.json", "traversal")
    if data is None:
        return None, why
    for k, t in data.items():
        if k != "_meta" and k.replace(" ", "").startswith(kernel.replace(" ", "").rstrip(">")) and "hbm_bytes_fetch_x2" in t:
            return {"bytes": int(t["hbm_bytes_fetch_x2"]), "fetch_bytes_x2": int(2 * t["FETCH_SIZE"] * 1024), "write_bytes": int(t["WRITE_SIZE"] * 1024), "source": name}, None
    return None, f"profiles/{name} holds no pass of kernel {kernel}"


def binding_bounds(kernel, ray_set, rays, steps_per_ray, kernel_ms, lds_steps_per_ray=0.0):
    """The bounds that bind the traversal kernel, against peaks MEASURED on this chip (profiles/rNN_calibration.json):
    node / triangle fetches per ns through the vector-memory pipeline (live: oracle visit counts x rays / HIP-event kernel
    time; `lds_steps_per_ray` of them are served from the LDS image of the default mapping instead and are reported against
    the LDS rate) and VALU wave-instructions per us per SIMD (instruction count from the committed SQ counter pass of THIS
    kernel on THESE sources / live kernel time)."""
    cal = latest_json("r*_calibration.json")
    if not cal:
        return None
    cal_name, cal = cal
    c, stale = kernel_counters(kernel, ray_set)
    fetches_per_ns = (steps_per_ray - lds_steps_per_ray) * rays / (kernel_ms * 1e6)
    if ray_set == "primary":
        peak, peak_kind = cal["node_fetch_peak_coherent"], "64-byte node per lane, neighbouring lanes share nodes, L1/L2-resident (vmem_peak 'coherent')"
    else:
        # incoherent rays: every lane its own node; blend of the scattered-L2 and scattered-MALL rates by the measured L2 hit rate
        hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else 0.85
        peak = 1.0 / (hit / cal["node_fetch_peak_scattered_l2"] + (1.0 - hit) / cal["node_fetch_peak_scattered_mall"])
        peak_kind = f"64-byte node per lane, scattered: L2 rate x {hit:.3f} + MALL rate x {1 - hit:.3f} ({'measured' if 'TCC_HIT_sum' in c else 'assumed'} TCC hit rate; vmem_peak 'scattered')"
    out = {"vmem_node_fetch": {"bound": "vector-memory pipeline (TA/L1/L2), node fetches", "unit": "fetches/ns", "achieved": round(fetches_per_ns, 2), "peak": round(peak, 2),
                               "frac": round(fetches_per_ns / peak, 4), "peak_kind": peak_kind, "steps_per_ray": round(steps_per_ray - lds_steps_per_ray, 3), "peak_source": cal_name}}
    if lds_steps_per_ray:
        # MI355X_MICROARCH.md, LDS table: ds_read_b128 4 and ds_read_b64 2 LDS cycles per wave-instruction when conflict-free -> 3 x 4 + 2 per 64 node records
        lds_peak = LDS_CLOCK_GHZ * cal["cus"] * 64 / 14.0
        lds_per_ns = lds_steps_per_ray * rays / (kernel_ms * 1e6)
        out["lds_fetch"] = {"bound": "LDS (top-of-tree image: 3 x ds_read_b128 + ds_read_b64 per node)", "unit": "fetches/ns", "achieved": round(lds_per_ns, 2),
                            "peak": round(lds_peak, 1), "frac": round(lds_per_ns / lds_peak, 4), "steps_per_ray": round(lds_steps_per_ray, 3),
                            "peak_kind": "conflict-free rate of the guide's LDS table at 2.4 GHz; distinct records on one bank quarter serialise"}
    if stale:
        out["counters_not_quoted"] = stale
    if "SQ_INSTS_VALU" in c:
        per_simd_us = c["SQ_INSTS_VALU"] / (kernel_ms * 1e3) / cal["simds"]
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        peak = cal["valu_issue_guide_2_cycle_rate"]           # MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD, at the clock measured in the calibration loop
        out["valu_issue"] = {"bound": "VALU issue", "unit": "wave-instructions/us/SIMD", "achieved": round(per_simd_us, 1), "peak": peak,
                             "frac": round(per_simd_us / peak, 4), "valu_instructions_per_launch": int(c["SQ_INSTS_VALU"]),
                             "lane_utilisation": round(lane_util, 4), "useful_lane_frac": round(per_simd_us / peak * lane_util, 4),
                             # scripts/ubench/valu_rate.hip: the same rate against the measured ceiling of the loop's own instruction classes
                             "frac_of_measured_loop_mix_ceiling": round(per_simd_us / cal["valu_issue_peak_loop_mix_r04"], 4), "measured_loop_mix_ceiling": cal["valu_issue_peak_loop_mix_r04"],
                             "ceilings": "peak = the guide's 2 cycles per wave64 instruction at the clock measured in the calibration loop (2 323 MHz -> 1 162); it holds for mul / add / fma with <= 2 VGPR "
                                         "sources only -- min / max / cndmask / compares / 3-source fma issue every ~4 cycles, so the loop's own mix tops out at 645 (profiles/r04_ubench_valu_rate.txt)",
                             "peak_source": cal_name, "counter_source": c.get("source")}
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
        out["ta_busy_frac_profiled"] = round(c["TA_TA_BUSY_sum"] / cal["cus"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)     # mean over the 256 TAs / cycles of one XCD
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out["wave_cycles_waiting_frac_profiled"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # wavefront occupancy: SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md), GRBM_GUI_ACTIVE the busy cycles of the
        # eight XCDs -> resident waves per SIMD averaged over the launch, against the 8 the hardware holds (the kernel's 64 VGPRs and
        # 80 KB of LDS per 16-wave workgroup admit all 8: what is missing from 8 is the launch's fill and drain)
        waves = 4.0 * c["SQ_WAVE_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8.0) * cal["simds"])
        out["occupancy"] = {"resident_waves_per_simd_time_averaged_profiled": round(waves, 2), "max_waves_per_simd": 8, "frac": round(waves / 8.0, 4),
                            "waves_per_simd_the_kernel_admits": 8}
    return out


def pick_bound(binding):
    """The top-level roofline: VALU issue -- whenever the committed counter pass belongs to the running sources; the live
    node-fetch bound stands in (and says so) while it does not."""
    for key in ("valu_issue", "vmem_node_fetch"):
        b = (binding or {}).get(key)
        if b:
            return key, b
    return None


# ------------------------------------------------------------------------------------------------
# timing
# ------------------------------------------------------------------------------------------------
def time_passes(abi, torch, bvh, rays_dev, hits_dev, n, variant, steps, warmup, dist, any_hit=False):
    """W untimed + K timed launches, back to back on one stream.  Returns (wall seconds for the K steps [max over ranks is taken by the caller],
    average launch duration in ms from ONE pair of HIP events around the timed region on the launch stream, ... and, from a second, untimed pass with an
    event pair around every single launch, the median and minimum of those).  Until round 4 the timed region itself carried an event pair per step:
    two marker packets between every two kernels cost 17 us per 0.18 ms step (profiles/r05_host_call_costs.txt: 174.9 us per launch back to back
    against 192.3 with them) -- time the benchmark spent measuring itself."""
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    first.record(stream)
    for i in range(steps):
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
    last.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    region_ms = first.elapsed_time(last) / max(1, steps)
    # per-launch spread (not part of the timed region): every launch between its own two events, which adds the dispatch latency the back-to-back region hides
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for i in range(steps):
        starts[i].record(stream)
        abi.traverse_async(bvh, rays_dev, hits_dev, n, any_hit, variant, stream)
        ends[i].record(stream)
    torch.cuda.synchronize()
    single = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    return wall, float(region_ms), float(np.median(single)), float(np.min(single))


def _collective_device(dist, dev):
    return "cpu" if dist.get_backend() == "gloo" else f"cuda:{dev}"


def max_over_ranks(torch, dist, dev, values):
    if dist is None:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device(dist, dev))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def gather_scalars(torch, dist, dev, values):
    """[[values of rank 0], [values of rank 1], ...] on every rank (bookkeeping, after the timed regions)."""
    if dist is None:
        return [list(values)]
    t = torch.tensor(list(values), dtype=torch.float64, device=_collective_device(dist, dev))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in g] for g in out]


# ------------------------------------------------------------------------------------------------
# renderer section (BASELINE configs 4 and 5)
# ------------------------------------------------------------------------------------------------
RENDER_CONFIGS = {
    # name: (scene, width, height, spp, max path length)
    "cfg4_cornell_1920x1080_64spp_len4": ("cornell", 1920, 1080, 64, 4),
    "cfg5_atrium_3840x2160_256spp_len8": ("atrium", 3840, 2160, 256, 8),
}


def scene_file(scene_name):
    """.rscene of a benchmark scene (converted once by rank 0; rodent_amd/bin/converter = the reference's converter as a
    table emitter)."""
    from rodent_amd import scene as S, scenes
    obj = scenes.scene_obj(scene_name)                                # (generates data/<scene>.obj where it has to)
    out = scenes.DATA / f"{scene_name.replace('/', '-d')}.bench.rscene"
    if not out.exists():
        scenes.DATA.mkdir(parents=True, exist_ok=True)
        S.convert(obj, out)
    return obj, out


def render_profile(config_name):
    """Per-kernel figures of the committed renderer profile of this configuration (profiles/rNN_render_profile_<cfg>.json,
    scripts/render_profile.sh: rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes of the
    same command), or the reason why none is quoted."""
    short = config_name.split("_")[0]
    name, data, why = current_profile(f"r*_render_profile_{short}.json", "render")
    if data is None:
        return {"not_quoted": why}
    out = {"source": name, "command": data.get("_meta", {}).get("command")}
    for mapping, kernels in data.items():
        if mapping == "_meta":
            continue
        rows = {}
        for k, v in kernels.items():
            row = {"calls_per_frame": v.get("calls_per_frame"), "avg_ms": round(v["avg_us"] / 1e3, 4), "ms_per_frame": round(v["avg_us"] * v.get("calls_per_frame", 0) / 1e3, 3)}
            if "hbm_TBps_fetch_x2" in v:
                row["hbm_frac"] = round(v["hbm_TBps_fetch_x2"] * 1e3 / HBM_PEAK_GBPS, 4)
                row["hbm_GBps"] = round(v["hbm_TBps_fetch_x2"] * 1e3, 1)
            rows[k] = row
        out[mapping] = rows
    return out


def render_section(args, torch, dist, rank, world, dev):
    """Frame rates of BASELINE configs 4 and 5.  One GPU: whole frames, every mapping.  N GPUs: config 5 only, row bands
    (parallel.row_band) + one film gather to rank 0, timed separately."""
    from rodent_amd import parallel, render as R, scene as S, scenes
    out = {}
    for name, (scene_name, w, h, spp, max_len) in RENDER_CONFIGS.items():
        if world > 1 and not name.startswith("cfg5"):
            continue
        if name.startswith("cfg5"):
            spp = args.render_spp5
        if rank == 0:
            scene_file(scene_name)
        if dist is not None:
            dist.barrier()
        obj, rscene = scene_file(scene_name)
        sc = S.Scene(rscene)
        eye, d, up, fov = scenes.CAMERAS[scene_name]
        cam = S.camera_settings(eye, d, up, fov, w, h)
        # N GPUs: rank r renders the interleaved 16-row tiles r, r + N, ... (bands of the atrium frame differ by 27 % in cost, tile shares by 1 %: profiles/r04_band_costs.txt)
        my_rows = sum(b - a for a, b in parallel.row_tiles(h, rank, world)) if world > 1 else h

        def render_share(r, it):
            if world > 1:
                r.render_tiles(cam, it, parallel.TILE_ROWS, rank, world)
            else:
                r.render_rows(cam, it, 0, h)
        frames = 3 if spp * w * h < (1 << 28) else 1
        entry = {"scene": f"{scene_name} ({sc.num_tris} triangles, {len(sc.materials)} materials)", "width": w, "height": h, "spp": spp, "max_path_len": max_len,
                 "samples_per_frame": spp * w * h, "timed_frames": frames, "rows_per_gpu": my_rows,
                 "partition": f"interleaved {parallel.TILE_ROWS}-row tiles" if world > 1 else "whole frame"}
        # auto = what the library chooses for this scene; streaming = the wavefront loop with the library's defaults (shading in stream
        # order); streaming_sorted = the same with the reference's sort by material in front of the shader (rodent_hip_render_sort)
        mappings = ["auto", "streaming", "streaming_sorted", "megakernel"]
        chosen = None
        for mapping in mappings:
            if world > 1 and mapping != "auto":                        # N GPUs: only the mapping the library chooses
                continue
            r = R.Renderer(sc, w, h, spp=4, max_path_len=max_len, dev=dev, mapping=mapping.split("_")[0], sort=True if mapping.endswith("_sorted") else None)
            if mapping == "auto":
                chosen = r.mapping_name()
                entry["auto_trace_refill_idle_lanes[bounce,shadow]"] = list(r.trace_refill())      # lane refill in the persistent traversal launches (0 = whole chunks)
            elif mapping == chosen:
                r.close()
                entry[mapping] = {"same_as": "auto"}
                continue
            render_share(r, 0)                                         # warm-up at 4 spp (allocations, code upload)
            r.configure(spp, max_len)
            r.clear()
            secs = []
            for it in range(frames):
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                render_share(r, it)                                    # synchronous: the share is in the device film when it returns
                secs.append(time.perf_counter() - t0)
            secs = [max_over_ranks(torch, dist, dev, [s])[0] for s in secs]
            best = float(np.median(secs))
            res = {"Msamples_s": round(spp * w * h / best / 1e6, 2), "frame_ms": round(best * 1e3, 2), "frame_ms_all": [round(s * 1e3, 2) for s in secs], "rays": r.counters()}
            if world > 1:
                # the one collective of the path: every peer's rows into rank 0's device film, then the frame is complete there
                film = parallel.device_film(dev)
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                full = parallel.gather_film_to_root(film, dist, tile_rows=parallel.TILE_ROWS)
                torch.cuda.synchronize()
                g = max_over_ranks(torch, dist, dev, [time.perf_counter() - t0])[0]
                res["film_gather_ms"] = round(g * 1e3, 3)
                res["film_gather_MB"] = round((h - my_rows) * w * 12 / 1e6, 2) if rank == 0 else None
                res["Msamples_s_including_gather"] = round(spp * w * h / (best + g) / 1e6, 2)
                if rank == 0:
                    res["film_complete_on_root"] = bool(torch.isfinite(full).all() and float(full[h - 1].abs().sum()) > 0 and float(full[0].abs().sum()) > 0)
            r.close()
            entry[mapping] = res
            if mapping == "auto":
                entry["auto_mapping"] = chosen
        entry["per_kernel_profiled"] = render_profile(name)
        out[name] = entry
    return out


def render_cpu_baseline(threads):
    """The reference's CPU mapping restated (oracle/cpu_wavefront.inc: tile-parallel wavefront renderer, hybrid ray8 x BVH8
    traversal, scalar shading) on this host: a bounded sample of each configuration (same scene, camera and path length,
    fewer pixels and samples per pixel)."""
    from oracle import binding as O
    from rodent_amd import formats as F, scene as S, scenes
    out = {}
    for name, (scene_name, w, h, spp, max_len) in RENDER_CONFIGS.items():
        obj, rscene = scene_file(scene_name)
        sc = S.Scene(rscene)
        n8, t8 = F.read_bvh(scenes.scene_bvh(scene_name), F.BVH8_TRI4)      # the reference's CPU targets trace a BVH8 / Tri4 (converter.cpp:152-259)
        # bounded samples (seconds, not minutes, of CPU work): config 4 whole (133 M samples), config 5 at a quarter of the pixels and 8 spp
        sw, sh, sspp = (w, h, spp) if scene_name == "cornell" else (w // 2, h // 2, 8)
        eye, d, up, fov = scenes.CAMERAS[scene_name]
        cam = S.camera_settings(eye, d, up, fov, sw, sh)
        O.render_wavefront(sc, n8, t8, cam, 0, 1, max_len, sw, sh, None, threads=threads)          # warm-up (thread start, page faults)
        t0 = time.perf_counter()
        O.render_wavefront(sc, n8, t8, cam, 0, sspp, max_len, sw, sh, None, threads=threads)
        dt = time.perf_counter() - t0
        out[name] = {"Msamples_s": round(sspp * sw * sh / dt / 1e6, 2), "cores": threads, "kind": "port",
                     "sample": f"{sw}x{sh}, {sspp} spp, path length {max_len}: {sspp * sw * sh} samples in {dt:.2f} s; reference's CPU wavefront mapping restated "
                               "(render/mapping_cpu.impala:352-473; scalar shading instead of RV-vectorised)"}
    return out


# ------------------------------------------------------------------------------------------------
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
        # RODENT_BENCH_BACKEND=gloo RODENT_BENCH_SHARE_GPUS=1 (test mode: the N-rank code path on a box with fewer GPUs than ranks --
        # ranks share devices, collectives go through the host); the driver's runs use RCCL, one rank per GPU
        backend = os.environ.get("RODENT_BENCH_BACKEND", "nccl")
        if os.environ.get("RODENT_BENCH_SHARE_GPUS", "0") not in ("", "0"):
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP traversal has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = local_rank
    info = not args.no_cpu_baseline and args.only is None          # the informational extras (never in the profiling runs)

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

    def ray_sets(sample, samples):
        if scene == "sponza":
            return (F.read_rays(scenes.DATA / "sponza-primary.rays", 0.0, scenes.PRIMARY_TMAX), F.read_rays(scenes.DATA / "sponza-random.rays", 0.0, scenes.RANDOM_TMAX))
        eye, d, up, fov = scenes.CAMERAS[scene]
        p = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, scenes.PRIMARY_TMAX, sample=sample, num_samples=samples)
        n4, _ = F.read_bvh(bvh_path, F.BVH4_TRI4)
        lo, hi = raygen.scene_bounds(n4)
        return p, raygen.random_rays(lo, hi, 1 << 20, 42 + sample, 0.0, scenes.RANDOM_TMAX)

    steps_p, warm_p = (args.steps, args.warmup) if args.only != "random" else (1, 0)
    steps_r, warm_r = (args.steps, args.warmup) if args.only != "primary" else (1, 0)

    def run_partition(prim, rnd):
        """Times both sets on this rank's share; returns the per-partition record (times already max over ranks)."""
        pd, rd = abi.to_device(prim, dev), abi.to_device(rnd, dev)
        hp = torch.zeros(max(len(prim), 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
        hr = torch.zeros(max(len(rnd), 1) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}")
        wall, k_mean, k_med, k_min = time_passes(abi, torch, bvh, pd, hp, len(prim), variant, steps_p, warm_p, dist)
        wall_r, kr_mean, kr_med, kr_min = time_passes(abi, torch, bvh, rd, hr, len(rnd), variant, steps_r, warm_r, dist)
        abi.check_errors(dev)
        wall, wall_r = max_over_ranks(torch, dist, dev, [wall, wall_r])
        per_rank = gather_scalars(torch, dist, dev, [k_mean, kr_mean, float(len(prim)), float(len(rnd))])
        total_p, total_r = int(sum(g[2] for g in per_rank)), int(sum(g[3] for g in per_rank))
        return {"prim": prim, "rnd": rnd, "prim_dev": pd, "rnd_dev": rd, "hits_dev": hp, "hits_rnd_dev": hr, "wall": wall, "wall_r": wall_r,
                "k": (k_mean, k_med, k_min), "kr": (kr_mean, kr_med, kr_min), "total": total_p, "total_rnd": total_r,
                "value": total_p * steps_p / wall / 1e6, "value_rnd": total_r * steps_r / wall_r / 1e6,
                "kernel_ms_per_rank": [[round(g[0], 5), round(g[1], 5)] for g in per_rank]}

    # ---- timed regions ------------------------------------------------------------------------
    prim_all, rnd_all = ray_sets(0, 1)
    strong_check = weak = None
    if world == 1:
        main_part = run_partition(prim_all, rnd_all)
        scaling = "strong"                                        # (N = 1 of the fixed 1 Mi-ray workload: the same set the N > 1 runs split)
    else:
        # strong: ONE ray set in contiguous ranges -- contiguous keeps coherent rays coherent
        a, b = parallel.ray_range(len(prim_all), rank, world)
        strong = run_partition(prim_all[a:b], rnd_all[a:b])
        # after the timed region: ONE gather of the Hit1 ranges to rank 0 (16 B/ray: 16 MiB in total), compared there with a
        # single-GPU trace of the whole set
        full = parallel.gather_hits_device(strong["hits_dev"], len(prim_all), dist, dev)
        full_rnd = parallel.gather_hits_device(strong["hits_rnd_dev"], len(rnd_all), dist, dev)
        if rank == 0:
            whole = abi.traverse(bvh, prim_all, variant=variant)
            whole_rnd = abi.traverse(bvh, rnd_all, variant=variant)
            strong_check = {"primary_equal_to_single_gpu": bool(full.tobytes() == whole.tobytes()), "random_equal_to_single_gpu": bool(full_rnd.tobytes() == whole_rnd.tobytes())}
        # weak: every rank its own 1 Mi rays (sub-pixel sample `rank` of `world`)
        pw, rw = ray_sets(rank, world)
        weak_part = run_partition(pw, rw)
        counts = gather_scalars(torch, dist, dev, [float((abi.from_device(weak_part["hits_dev"], F.HIT1)["tri_id"] >= 0).sum()),
                                                   float((abi.from_device(weak_part["hits_rnd_dev"], F.HIT1)["tri_id"] >= 0).sum())])
        weak = {"Mrays_s": round(weak_part["value"], 3), "ms_per_step": round(1e3 * weak_part["wall"] / steps_p, 5), "random_Mrays_s": round(weak_part["value_rnd"], 3),
                "rays_per_gpu_per_step": len(pw), "kernel_ms_per_rank[primary,random]": weak_part["kernel_ms_per_rank"],
                "hit_counts_per_rank[primary,random]": [[int(c[0]), int(c[1])] for c in counts],
                "what": "rank r traces sub-pixel sample r of N through the same 1024 x 1024 pixel grid (random: seed 42 + r): per-GPU work fixed"}
        strong_rec = {"Mrays_s": round(strong["value"], 3), "ms_per_step": round(1e3 * strong["wall"] / steps_p, 5), "random_Mrays_s": round(strong["value_rnd"], 3),
                      "rays_per_gpu_per_step": len(strong["prim"]), "kernel_ms_per_rank[primary,random]": strong["kernel_ms_per_rank"],
                      "what": "ONE 1 Mi-ray set in contiguous ranges (SURVEY 8e), Hit1 gather to rank 0 after the timed region"}
        main_part, scaling = (weak_part, "weak") if args.weak and not args.strong else (strong, "strong")
    prim, rnd = main_part["prim"], main_part["rnd"]
    n = len(prim)
    prim_dev, rnd_dev, hits_dev, hits_rnd_dev = main_part["prim_dev"], main_part["rnd_dev"], main_part["hits_dev"], main_part["hits_rnd_dev"]
    k_mean, k_med, k_min = main_part["k"]
    kr_mean, kr_med, kr_min = main_part["kr"]

    # the same two sets with the schedule history on (rodent_hip_schedule_history: chunks traced longest first by the previous
    # launch's per-chunk cost -- state carried from step to step, therefore NOT the headline; same hit records)
    history = None
    if width == 2 and abi.variants(2)[variant] == "top" and world == 1 and info:
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
    sorted_rec = None
    if world == 1 and "sorted" in abi.variants(width) and args.only != "primary":
        sv = abi.variants(width).index("sorted")
        hits_sorted_dev = torch.zeros_like(hits_rnd_dev)
        wall_s, ks_mean, _, _ = time_passes(abi, torch, bvh, rnd_dev, hits_sorted_dev, len(rnd), sv, steps_r, warm_r, None)
        sorted_rec = {"Mrays_s": round(len(rnd) * steps_r / wall_s / 1e6, 3), "ms_per_step": round(1e3 * wall_s / steps_r, 5), "kernels_ms": round(ks_mean, 5),
                      "variant": "sorted: counting sort of the rays on 512 Morton cells of their origin inside every launch, then the default kernel through the permutation",
                      "identical_to_unsorted": bool(torch.equal(hits_sorted_dev, hits_rnd_dev))}
    # ... and through "refill": continuous compaction inside the persistent kernel (a wave replaces finished rays instead of waiting for
    # the last ray of a 64-ray chunk) -- the mapping for incoherent ray sets; the renderer's bounce and shadow passes use the same scheme
    refill_rec = None
    if world == 1 and "refill" in abi.variants(width) and args.only != "primary":
        rv = abi.variants(width).index("refill")
        hits_refill_dev = torch.zeros_like(hits_rnd_dev)
        wall_f, kf_mean, _, _ = time_passes(abi, torch, bvh, rnd_dev, hits_refill_dev, len(rnd), rv, steps_r, warm_r, None)
        refill_rec = {"Mrays_s": round(len(rnd) * steps_r / wall_f / 1e6, 3), "ms_per_step": round(1e3 * wall_f / steps_r, 5), "kernels_ms": round(kf_mean, 5),
                      "variant": "refill: the default's persistent workgroups; a wave whose idle lanes reach 32 draws that many new rays from its stripe's counter",
                      "identical_to_default": bool(torch.equal(hits_refill_dev, hits_rnd_dev))}
    # the default WITH the ray-kind hint (rodent_hip_ray_kind_hint(1); off by default from round 5 on): state carried from launch to launch, therefore not the headline
    hint_rec = None
    if world == 1 and width == 2 and abi.variants(2)[variant] == "top" and args.only != "primary":
        abi.ray_kind_hint(True)
        hits_hint_dev = torch.zeros_like(hits_rnd_dev)
        wall_n, kn_mean, _, _ = time_passes(abi, torch, bvh, rnd_dev, hits_hint_dev, len(rnd), variant, steps_r, warm_r, None)
        abi.ray_kind_hint(False)
        hint_rec = {"Mrays_s": round(len(rnd) * steps_r / wall_n / 1e6, 3), "kernels_ms": round(kn_mean, 5), "identical_to_default": bool(torch.equal(hits_hint_dev, hits_rnd_dev)),
                    "what": "rodent_hip_ray_kind_hint(1): the list goes to k_bvh2_top_refill from its second launch on; the default (random_Mrays_s) is k_bvh2_top_auto alone, whose waves "
                            "find their rays incoherent and run the refill loop -- no state between launches"}
    # the default WITHOUT the tile mapping (rodent_hip_ray_grid(0): camera rays traced in list order, 64 pixels of a row per wavefront, as until round 4)
    list_order_rec = None
    if world == 1 and width == 2 and abi.variants(2)[variant] == "top" and args.only != "random" and not args.no_cpu_baseline:      # (not in the profiling runs: the same kernel name in another mode would mix into their per-kernel means)
        abi.ray_grid(0)
        hits_lo_dev = torch.zeros_like(hits_dev)
        wall_l, kl_mean, _, _ = time_passes(abi, torch, bvh, prim_dev, hits_lo_dev, n, variant, steps_p, warm_p, None)
        abi.ray_grid(-1)
        list_order_rec = {"Mrays_s": round(n * steps_p / wall_l / 1e6, 3), "kernels_ms": round(kl_mean, 5), "identical_to_default": bool(torch.equal(hits_lo_dev, hits_dev)),
                          "what": "rodent_hip_ray_grid(0): the same kernel with the wave's 64 rays in list order (64 pixels of an image row); the default recognises the image "
                                  "width from 66 of the launch's rays and gives every wavefront an 8 x 8-pixel tile -- no state between launches, hit records identical"}
    abi.check_errors(dev)                                         # the asynchronous entry points report stack overflows through a flag
    # for information only (never `value`): independent batches in flight on two streams -- the fill of one launch
    # overlaps the drain of the other (every (device, stream) has its own launch state)
    overlapped = big = big_random = None
    if info and world == 1:
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
        # the same camera at 4096 x 4096 = 16 Mi primary rays per launch -- the kernel's throughput regime (at 1 Mi rays
        # the launch is bound by its schedule, LAB_NOTES.md 3.1.1)
        if scene != "sponza":
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
            # ... and 8 Mi random segments per launch (what a renderer's bounce pass looks like): the default mapping and "refill"
            try:
                if width == 2 and "refill" in abi.variants(width):
                    lo8, hi8 = raygen.scene_bounds(F.read_bvh(bvh_path, F.BVH4_TRI4)[0])
                    rnd8 = raygen.random_rays(lo8, hi8, 1 << 23, 43, 0.0, scenes.RANDOM_TMAX)
                    rnd8_dev = abi.to_device(rnd8, dev)
                    h8 = [torch.zeros(len(rnd8) * F.HIT1.itemsize, dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(2)]
                    wall_d, kd, _, _ = time_passes(abi, torch, bvh, rnd8_dev, h8[0], len(rnd8), variant, 10, 3, None)
                    wall_f8, kf8, _, _ = time_passes(abi, torch, bvh, rnd8_dev, h8[1], len(rnd8), abi.variants(width).index("refill"), 10, 3, None)
                    big_random = {"rays_per_launch": len(rnd8), "default_Mrays_s": round(len(rnd8) * 10 / wall_d / 1e6, 3), "default_kernels_ms": round(kd, 5),
                                  "refill_Mrays_s": round(len(rnd8) * 10 / wall_f8 / 1e6, 3), "refill_kernels_ms": round(kf8, 5), "identical_hits": bool(torch.equal(h8[0], h8[1]))}
                    del rnd8_dev, h8, rnd8
            except Exception as e:
                print(f"bench.py: 8 Mi random-ray measurement skipped ({e})", file=sys.stderr)

    hits = abi.from_device(hits_dev, F.HIT1)[:n]
    hits_rnd = abi.from_device(hits_rnd_dev, F.HIT1)[:len(rnd)]

    # ---- the other scene classes and the any-hit ray class (scripts/scene_matrix.py; the oracle checks a 32 Ki-ray sample of every cell) ----
    scene_rows = None
    if info and world == 1 and width == 2 and not args.no_scenes and scene != "sponza":
        try:
            sys.path.insert(0, str(ROOT / "scripts"))
            import scene_matrix                                          # lab tooling; imports the oracle as its checker
            scene_rows, t_scenes = {}, time.time()
            for sc_name in (scene, "gallery", "crown", "plant"):
                if time.time() - t_scenes > 300:                         # a slow host must not cost the bench line
                    scene_rows[sc_name] = {"skipped": "the scenes before this one took more than 300 s to build and trace"}
                    continue
                scene_rows[sc_name] = scene_matrix.measure(sc_name, steps=args.steps, quiet=True)
        except Exception as e:                                          # informational: never lose the bench line over it
            print(f"bench.py: scene matrix skipped ({e})", file=sys.stderr)
            scene_rows = scene_rows or None

    # ---- renderer (BASELINE configs 4 / 5): after the traversal's timed regions, own timed frames ----
    render = None
    if info and not args.no_render and width == 2:
        render = render_section(args, torch, dist, rank, world, dev)

    if rank != 0:
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    # ---- rank 0: the JSON line -----------------------------------------------------------------
    kname = abi.kernel_name(width, variant)
    sharding = {"strong": "ONE ray set in contiguous ranges (strong scaling), Hit1 gather to rank 0 after the timed region", "weak": "rays sharded by sub-pixel sample (weak scaling)"}[scaling]
    out = {
        "metric": "Mrays/s", "value": round(main_part["value"], 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * main_part["wall"] / steps_p, 5), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene}.bvh + {scene}-primary.rays (1024x1024 primary rays, tmax 5000, closest hit)" + (" per GPU" if scaling == "weak" and world > 1 else ""),
                   "rays_per_gpu_per_step": n, "bvh_layout": f"BVH{width}", "kernel": kname,
                   "variant": abi.variants(width)[variant], "parallelism": f"replicated BVH x {world}" + (f", {sharding}" if world > 1 else ""),
                   "world_size": world, "collective_backend": (None if dist is None else ("nccl (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend() + " (test mode)"))},
        "extra": {"random_Mrays_s": round(main_part["value_rnd"], 3), "random_ms_per_step": round(1e3 * main_part["wall_r"] / steps_r, 5),
                  # mean: HIP events around the whole timed region / steps; single_*: each launch between its own event pair in a second pass (adds the dispatch latency)
                  "primary_kernel_ms": {"mean": round(k_mean, 5), "single_launch_median": round(k_med, 5), "single_launch_min": round(k_min, 5)},
                  "random_kernel_ms": {"mean": round(kr_mean, 5), "single_launch_median": round(kr_med, 5), "single_launch_min": round(kr_min, 5)},
                  "kernel_ms_per_rank[primary,random]": main_part["kernel_ms_per_rank"],
                  "hit_counts[primary,random]": [int((hits["tri_id"] >= 0).sum()), int((hits_rnd["tri_id"] >= 0).sum())],
                  "library": {"version": abi.lib().rodent_hip_version().decode(), "source_digest": abi.lib().rodent_hip_source_digest().decode(),
                              "built_from_these_sources": abi.built_from_these_sources()},       # the prebuilt .so against the sources lying next to it
                  "two_streams_Mrays_s_per_gpu": None if overlapped is None else round(overlapped, 3),
                  "primary_16Mi_rays_per_launch": big, "random_8Mi_rays_per_launch": big_random},
    }
    if world > 1:
        out["extra"]["strong_scaling"] = strong_rec
        out["extra"]["weak_scaling"] = weak
        out["extra"]["strong_scaling_check"] = strong_check
    if history is not None:
        out["extra"]["with_schedule_history"] = history
    if sorted_rec is not None:
        out["extra"]["random_sorted"] = sorted_rec
    if refill_rec:
        out["extra"]["random_refill"] = refill_rec
    if hint_rec:
        out["extra"]["random_with_kind_hint"] = hint_rec
    if render is not None:
        out["extra"]["render"] = render
    if scene_rows:
        out["extra"]["scenes"] = scene_rows
        out["extra"]["scenes_what"] = ("1 Mi camera rays / random segments (closest hit) / ao rays (ray_gen shadow, any hit, tmax 0.999) per scene class through the default mapping (top) and "
                                       "'fast' / 'refill' beside it: kernel ms, Mrays/s of the default, oracle parity of a 32 Ki-ray sample, oracle visits per ray and stack depths, blocks spilled")
        out["config"]["ao_Mrays_s"] = (scene_rows.get(scene, {}).get("ao") or {}).get("Mrays_s")
        out["config"]["scene_classes_Mrays_s[primary,random,ao]"] = {k: [(v.get(c) or {}).get("Mrays_s") for c in ("primary", "random", "ao")] for k, v in scene_rows.items() if "primary" in v}
        out["config"]["scene_classes_parity"] = all(v[c]["sample_parity"] for v in scene_rows.values() if "primary" in v for c in ("primary", "random", "ao"))
    # what the driver's record keeps is `config`, `roofline` and `cpu_baseline`: the other headline figures as short scalars
    out["config"]["random_Mrays_s"] = round(main_part["value_rnd"], 1)          # stateless: the in-kernel choice alone (the ray-kind hint is off by default)
    if hint_rec:
        out["config"]["random_with_kind_hint_Mrays_s"] = hint_rec["Mrays_s"]
    if list_order_rec:
        out["config"]["primary_in_list_order_Mrays_s"] = list_order_rec["Mrays_s"]
        out["extra"]["primary_in_list_order"] = list_order_rec
    if world > 1:
        # what one GPU predicts for N (profiles/r05_range_costs.txt, r04_band_costs.txt: every rank's share timed alone): ONE 1 Mi-ray set does not shard its tail
        out["config"]["predicted_scaling_x"] = {"strong_1Mi_primary_contiguous_ranges": {"2": 1.27, "4": 1.69, "8": 1.94}, "strong_1Mi_random": {"2": 1.33, "4": 1.74, "8": 1.97},
                                                "cfg5_frame_interleaved_16_row_tiles": {"2": 1.99, "4": 3.99, "8": 7.94}, "weak": "1 Mi rays per GPU: ~N",
                                                "source": "profiles/r05_range_costs.txt, profiles/r04_band_costs.txt"}
        out["config"]["world_size_seen_by_the_collective_backend"] = dist.get_world_size()
        out["config"]["strong_scaling_Mrays_s[primary,random]"] = [strong_rec["Mrays_s"], strong_rec["random_Mrays_s"]]     # ONE 1 Mi-ray set over the N GPUs
        out["config"]["weak_scaling_Mrays_s[primary,random]"] = [weak["Mrays_s"], weak["random_Mrays_s"]]
    if big:
        out["config"]["primary_16Mi_Mrays_s"] = big.get("Mrays_s")
    if big_random:
        out["config"]["random_8Mi_Mrays_s"] = big_random.get("default_Mrays_s")
    for cfg_name, entry in (render or {}).items():
        auto = entry.get("auto") or {}
        if "Msamples_s" in auto:
            out["config"][cfg_name.split("_")[0] + "_Msamples_s"] = auto["Msamples_s"]
            out["config"][cfg_name.split("_")[0] + "_mapping"] = entry.get("auto_mapping")
    if not args.no_cpu_baseline:
        from oracle import binding as O      # checker / CPU baseline only: never on the measured path
        # (1) visit counts of the reference algorithm for THIS layout over ALL rays -> algorithmic bytes per ray; full parity check
        block = {2: F.BVH2_TRI1, 4: F.BVH4_TRI4, 8: F.BVH8_TRI4}[width]
        nodes, tris = F.read_bvh(bvh_path, block)
        node_b, prim_b, algo = {2: (64, 48, "ref"), 4: (128, 224, "gpu"), 8: (256, 224, "gpu")}[width]
        ref_hits, st = O.traverse(width, nodes, tris, prim, algo=algo)
        ref_rnd, st_r = O.traverse(width, nodes, tris, rnd, algo=algo)
        lds_p = lds_r = 0.0
        if width == 2 and abi.variants(2)[variant] == "top" and n >= 6144 * 64:
            # the share of the node visits that the default mapping serves from its LDS image (host restatement of the image's node set)
            from rodent_amd import topimage
            ids = topimage.image_nodes(nodes)
            lds_p = float(O.node_visits(nodes, tris, prim)[ids].sum()) / len(prim)
            lds_r = float(O.node_visits(nodes, tris, rnd)[ids].sum()) / len(rnd)
        bytes_per_ray = 32 + 16 + node_b * st["inner_per_ray"] + prim_b * st["prims_per_ray"]
        achieved = bytes_per_ray * n / (k_mean * 1e-3) / 1e9
        traffic, traffic_why = measured_traffic(kname)
        binding = binding_bounds(kname, "primary", n, st["inner_per_ray"] + st["prims_per_ray"], k_mean, lds_p)
        kname_r = kname                               # (the ray-kind hint is off by default: both sets run through the same kernel)
        binding_r = binding_bounds(kname_r, "random", len(rnd), st_r["inner_per_ray"] + st_r["prims_per_ray"], kr_mean, lds_r)
        top = pick_bound(binding)
        roof = {"bound": top[0], "achieved": top[1]["achieved"], "peak": top[1]["peak"], "unit": top[1]["unit"], "frac": top[1]["frac"]} if top else \
               {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "note": "no calibration under profiles/"}
        compulsory = 48 * n + nodes.nbytes + tris.nbytes
        vi = (binding or {}).get("valu_issue") or {}
        hbm_alg = achieved / HBM_PEAK_GBPS
        roof.update({
            "traffic": None if traffic is None else traffic["bytes"],
            "kernel": kname, "kernel_ms": round(k_mean, 5),
            "frac_of_measured_loop_mix_ceiling": vi.get("frac_of_measured_loop_mix_ceiling"), "lane_utilisation": vi.get("lane_utilisation"),
            # BASELINE's "fraction of HBM roofline", one key: fabric bytes of the committed --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) / live kernel time / 8 TB/s, beside the
            # compulsory bytes (rays in, hits out, the BVH once) and the write amplification (WRITE_SIZE / the 16-byte Hit1 array)
            "hbm": {"measured_frac": None if traffic is None else round(traffic["bytes"] / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "compulsory_frac": round(compulsory / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "traffic_over_compulsory": None if traffic is None else round(traffic["bytes"] / compulsory, 3),
                    "write_amplification": None if traffic is None else round(traffic["write_bytes"] / (16.0 * n), 3),
                    "peak_GBps": HBM_PEAK_GBPS, "source": None if traffic is None else traffic["source"]},
            "hbm_measured_frac": None if traffic is None else round(traffic["bytes"] / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "cache_served_bytes_over_hbm_peak": round(hbm_alg, 4),
            "cache_served_bytes_over_hbm_peak_is": "NOT a fraction: SURVEY 8(d)'s bytes per ray count every node / triangle visit although the BVH is served by LDS / L1 / L2 / MALL (> 1 is expected)",
            "what": "VALU issue: wave-instructions per us per SIMD of this kernel (SQ_INSTS_VALU of the committed counter pass / live kernel time / SIMDs) against the guide's 2-cycle rate at the "
                    "measured clock.  At 1 Mi rays per launch the launch is a tail (LAB_NOTES.md 3.1.1); the same kernel at 16 Mi rays is in extra.primary_16Mi_rays_per_launch"
                    + ("" if top and top[0] == "valu_issue" else "  [the committed counter pass does not belong to the running sources: the live node-fetch bound stands in]"),
            "hbm_algorithmic": {"bound": "hbm", "GBps": round(achieved, 2), "peak_GBps": HBM_PEAK_GBPS, "frac": round(hbm_alg, 5), "bytes_per_ray": round(bytes_per_ray, 2),
                                "visits_per_ray": {"inner": round(st["inner_per_ray"], 3), "prim": round(st["prims_per_ray"], 3)},
                                "note": "SURVEY 8(d): 32 + 16 + 64 N_inner + 48 N_tri bytes per ray x rays / kernel time: a count of cache hits, not a fraction of anything"},
            # FETCH_SIZE x 2 = 128-byte line fills, calibrated on scattered 64-byte node fetches as well (profiles/r04_fetch_size_calibration.txt): bytes between the L2s and the fabric,
            # Infinity-Cache hits included -- an upper bound on DRAM bytes
            "l2_fabric_traffic": {"not_quoted": traffic_why} if traffic is None else
                            {"what": "FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes: bytes between the L2s and the fabric, Infinity-Cache hits included (an upper bound on HBM bytes)",
                             "bytes_per_launch": traffic["bytes"], "GBps": round(traffic["bytes"] / (k_mean * 1e-3) / 1e9, 1), "frac": round(traffic["bytes"] / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "compulsory_bytes_per_launch": int(compulsory), "over_compulsory": round(traffic["bytes"] / compulsory, 3),
                             "write_bytes": traffic["write_bytes"], "write_over_hit1_array": round(traffic["write_bytes"] / (16.0 * n), 3), "source": traffic["source"]},
            "binding": binding,
            "random": {"kernel": kname_r, "kernel_ms": round(kr_mean, 5), "visits_per_ray": {"inner": round(st_r["inner_per_ray"], 3), "prim": round(st_r["prims_per_ray"], 3)},
                       "bound": (pick_bound(binding_r) or [None])[0], "frac": (pick_bound(binding_r) or [None, {"frac": None}])[1]["frac"], "binding": binding_r}})
        out["roofline"] = roof
        out["config"]["hbm_measured_frac"] = roof["hbm_measured_frac"]
        out["config"]["cache_served_bytes_over_hbm_peak"] = roof["cache_served_bytes_over_hbm_peak"]
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
        if render is not None:
            try:
                out["extra"]["render"]["cpu_baseline"] = render_cpu_baseline(threads)
            except Exception as e:                                # informational: never lose the bench line over it
                out["extra"]["render"]["cpu_baseline"] = {"skipped": str(e)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
