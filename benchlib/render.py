"""Renderer section of bench.py (BASELINE configs 4 and 5 through the renderer ABI) and its CPU baseline."""
from __future__ import annotations

import json
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

from .profiles import render_profile
from .timing import max_over_ranks

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
        # N GPUs: rank r renders the interleaved 16-row tiles r, r + N, ... (bands of the atrium frame differ by 27 % in cost, tile shares
        # by 1 %: profiles/r04_band_costs.txt)
        my_rows = sum(b - a for a, b in parallel.row_tiles(h, rank, world)) if world > 1 else h

        def render_share(r, it):
            if world > 1:
                r.render_tiles(cam, it, parallel.TILE_ROWS, rank, world)
            else:
                r.render_rows(cam, it, 0, h)
        frames = 3 if spp * w * h < (1 << 28) else 1
        entry = {"scene": f"{scene_name} ({sc.num_tris} triangles, {len(sc.materials)} materials)", "width": w, "height": h, "spp": spp,
            "max_path_len": max_len,
                 "samples_per_frame": spp * w * h, "timed_frames": frames, "rows_per_gpu": my_rows,
                 "partition": f"interleaved {parallel.TILE_ROWS}-row tiles" if world > 1 else "whole frame"}
        # auto = what the library chooses for this scene; streaming = the wavefront loop with the library's defaults (shading in stream
        # order); streaming_sorted = the same with the reference's sort by material in front of the shader (rodent_hip_render_sort)
        mappings = ["auto", "streaming", "streaming_sorted", "megakernel"]
        chosen = None
        for mapping in mappings:
            if world > 1 and mapping != "auto":                        # N GPUs: only the mapping the library chooses
                continue
            r = R.Renderer(sc, w, h, spp=4, max_path_len=max_len, dev=dev, mapping=mapping.split("_")[0],
                sort=True if mapping.endswith("_sorted") else None)
            if mapping == "auto":
                chosen = r.mapping_name()
                # lane refill in the persistent traversal launches (0 = whole chunks)
                entry["auto_trace_refill_idle_lanes[bounce,shadow]"] = list(r.trace_refill())
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
            res = {"Msamples_s": round(spp * w * h / best / 1e6, 2), "frame_ms": round(best * 1e3, 2),
                "frame_ms_all": [round(s * 1e3, 2) for s in secs], "rays": r.counters()}
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
                    res["film_complete_on_root"] = bool(torch.isfinite(full).all() and float(full[h - 1].abs().sum()) > 0
                        and float(full[0].abs().sum()) > 0)
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
        # the reference's CPU targets trace a BVH8 / Tri4 (converter.cpp:152-259)
        n8, t8 = F.read_bvh(scenes.scene_bvh(scene_name), F.BVH8_TRI4)
        # bounded samples (seconds, not minutes, of CPU work): config 4 whole (133 M samples), config 5 at a quarter of the pixels and 8 spp
        sw, sh, sspp = (w, h, spp) if scene_name == "cornell" else (w // 2, h // 2, 8)
        eye, d, up, fov = scenes.CAMERAS[scene_name]
        cam = S.camera_settings(eye, d, up, fov, sw, sh)
        O.render_wavefront(sc, n8, t8, cam, 0, 1, max_len, sw, sh, None, threads=threads)          # warm-up (thread start, page faults)
        t0 = time.perf_counter()
        O.render_wavefront(sc, n8, t8, cam, 0, sspp, max_len, sw, sh, None, threads=threads)
        dt = time.perf_counter() - t0
        out[name] = {"Msamples_s": round(sspp * sw * sh / dt / 1e6, 2), "cores": threads, "kind": "port",
                     "sample": f"{sw}x{sh}, {sspp} spp, path length {max_len}: {sspp * sw * sh} samples in {dt:.2f} s; reference's CPU "
                         f"wavefront mapping restated "
                               "(render/mapping_cpu.impala:352-473; scalar shading instead of RV-vectorised)"}
    return out
