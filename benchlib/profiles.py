"""Committed profiles (profiles/rNN_*.json) and the rooflines bench.py derives from them.  Counter-derived figures are quoted only while the
profile carries the hash of the kernel sources that are running (rodent_amd/provenance.py)."""
from __future__ import annotations

import json
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LDS_CLOCK_GHZ = 2.4             # same guide: 2.4 GHz engine clock


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
        return name, None, f"profiles/{name} was taken on other {kind} sources (source_sha {data.get('_meta', {}).get('source_sha')} != " \
            f"{provenance.source_sha(kind)}): not quoted"
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
            return {"bytes": int(t["hbm_bytes_fetch_x2"]), "fetch_bytes_x2": int(2 * t["FETCH_SIZE"] * 1024),
                "write_bytes": int(t["WRITE_SIZE"] * 1024), "source": name}, None
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
        peak, peak_kind = cal["node_fetch_peak_coherent"], "64-byte node per lane, neighbouring lanes share nodes, L1/L2-resident " \
            "(vmem_peak 'coherent')"
    else:
        # incoherent rays: every lane its own node; blend of the scattered-L2 and scattered-MALL rates by the measured L2 hit rate
        hit = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else 0.85
        peak = 1.0 / (hit / cal["node_fetch_peak_scattered_l2"] + (1.0 - hit) / cal["node_fetch_peak_scattered_mall"])
        peak_kind = f"64-byte node per lane, scattered: L2 rate x {hit:.3f} + MALL rate x {1 - hit:.3f} " \
            f"({'measured' if 'TCC_HIT_sum' in c else 'assumed'} TCC hit rate; vmem_peak 'scattered')"
    out = {"vmem_node_fetch": {"bound": "vector-memory pipeline (TA/L1/L2), node fetches", "unit": "fetches/ns",
        "achieved": round(fetches_per_ns, 2), "peak": round(peak, 2),
                               "frac": round(fetches_per_ns / peak, 4), "peak_kind": peak_kind,
                                   "steps_per_ray": round(steps_per_ray - lds_steps_per_ray, 3), "peak_source": cal_name}}
    if lds_steps_per_ray:
        # MI355X_MICROARCH.md, LDS table: ds_read_b128 4 and ds_read_b64 2 LDS cycles per wave-instruction when conflict-free -> 3 x 4 + 2
        # per 64 node records
        lds_peak = LDS_CLOCK_GHZ * cal["cus"] * 64 / 14.0
        lds_per_ns = lds_steps_per_ray * rays / (kernel_ms * 1e6)
        out["lds_fetch"] = {"bound": "LDS (top-of-tree image: 3 x ds_read_b128 + ds_read_b64 per node)", "unit": "fetches/ns",
            "achieved": round(lds_per_ns, 2),
                            "peak": round(lds_peak, 1), "frac": round(lds_per_ns / lds_peak, 4),
                                "steps_per_ray": round(lds_steps_per_ray, 3),
                            "peak_kind": "conflict-free rate of the guide's LDS table at 2.4 GHz; distinct records on one bank quarter "
                                "serialise"}
    if stale:
        out["counters_not_quoted"] = stale
    if "SQ_INSTS_VALU" in c:
        per_simd_us = c["SQ_INSTS_VALU"] / (kernel_ms * 1e3) / cal["simds"]
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        # MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD, at the clock measured in the calibration loop
        peak = cal["valu_issue_guide_2_cycle_rate"]
        out["valu_issue"] = {"bound": "VALU issue", "unit": "wave-instructions/us/SIMD", "achieved": round(per_simd_us, 1), "peak": peak,
                             "frac": round(per_simd_us / peak, 4), "valu_instructions_per_launch": int(c["SQ_INSTS_VALU"]),
                             "lane_utilisation": round(lane_util, 4), "useful_lane_frac": round(per_simd_us / peak * lane_util, 4),
                             # scripts/ubench/valu_rate.hip: the same rate against the measured ceiling of the loop's own instruction
                             # classes
                             "frac_of_measured_loop_mix_ceiling": round(per_simd_us / cal["valu_issue_peak_loop_mix_r04"], 4),
                                 "measured_loop_mix_ceiling": cal["valu_issue_peak_loop_mix_r04"],
                             "ceilings": "peak = the guide's 2 cycles per wave64 instruction at the clock measured in the calibration loop "
                                 "(2 323 MHz -> 1 162); it holds for mul / add / fma with <= 2 VGPR "
                                         "sources only -- min / max / cndmask / compares / 3-source fma issue every ~4 cycles, so the "
                                             "loop's own mix tops out at 645 (profiles/r04_ubench_valu_rate.txt)",
                             "peak_source": cal_name, "counter_source": c.get("source")}
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
        # mean over the 256 TAs / cycles of one XCD
        out["ta_busy_frac_profiled"] = round(c["TA_TA_BUSY_sum"] / cal["cus"] / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        out["wave_cycles_waiting_frac_profiled"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # wavefront occupancy: SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md), GRBM_GUI_ACTIVE the busy cycles of
        # the eight XCDs -> resident waves per SIMD averaged over the launch, against the 8 the hardware holds (the kernel's 64 VGPRs and 80
        # KB of LDS per 16-wave workgroup admit all 8: what is missing from 8 is the launch's fill and drain)
        waves = 4.0 * c["SQ_WAVE_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8.0) * cal["simds"])
        out["occupancy"] = {"resident_waves_per_simd_time_averaged_profiled": round(waves, 2), "max_waves_per_simd": 8,
            "frac": round(waves / 8.0, 4),
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
            row = {"calls_per_frame": v.get("calls_per_frame"), "avg_ms": round(v["avg_us"] / 1e3, 4),
                "ms_per_frame": round(v["avg_us"] * v.get("calls_per_frame", 0) / 1e3, 3)}
            if "hbm_TBps_fetch_x2" in v:
                row["hbm_frac"] = round(v["hbm_TBps_fetch_x2"] * 1e3 / HBM_PEAK_GBPS, 4)
                row["hbm_GBps"] = round(v["hbm_TBps_fetch_x2"] * 1e3, 1)
            rows[k] = row
        out[mapping] = rows
    return out


def traversal_roofline(b, part, hits, hits_rnd, kname):
    """The `roofline` object of the bench line and the full-parity record: visit counts of the reference algorithm for THIS layout over
    ALL rays (oracle: checker only) -> algorithmic bytes per ray; the binding bounds from the committed counter passes; BASELINE's
    "fraction of HBM roofline" as `roofline.hbm`.  Returns (roofline, {"primary": bool, "random": bool})."""
    from oracle import binding as O      # checker only: never on the measured path
    from rodent_amd import formats as F
    prim, rnd, width, variant, abi = part["prim"], part["rnd"], b.width, b.variant, b.abi
    n, k_mean, kr_mean = len(prim), part["k"][0], part["kr"][0]
    block = {2: F.BVH2_TRI1, 4: F.BVH4_TRI4, 8: F.BVH8_TRI4}[width]
    nodes, tris = F.read_bvh(b.bvh_path, block)
    node_b, prim_b, algo = {2: (64, 48, "ref"), 4: (128, 224, "gpu"), 8: (256, 224, "gpu")}[width]
    ref_hits, st = O.traverse(width, nodes, tris, prim, algo=algo)
    ref_rnd, st_r = O.traverse(width, nodes, tris, rnd, algo=algo)
    lds_p = lds_r = 0.0
    if b.default_top and n >= 6144 * 64:
        # the share of the node visits that the default mapping serves from its LDS image (host restatement of the image's node set)
        from rodent_amd import topimage
        ids = topimage.image_nodes(nodes)
        lds_p = float(O.node_visits(nodes, tris, prim)[ids].sum()) / len(prim)
        lds_r = float(O.node_visits(nodes, tris, rnd)[ids].sum()) / len(rnd)
    bytes_per_ray = 32 + 16 + node_b * st["inner_per_ray"] + prim_b * st["prims_per_ray"]
    achieved = bytes_per_ray * n / (k_mean * 1e-3) / 1e9
    traffic, traffic_why = measured_traffic(kname)
    binding = binding_bounds(kname, "primary", n, st["inner_per_ray"] + st["prims_per_ray"], k_mean, lds_p)
    # (the ray-kind hint is off by default: both sets run through the same kernel)
    binding_r = binding_bounds(kname, "random", len(rnd), st_r["inner_per_ray"] + st_r["prims_per_ray"], kr_mean, lds_r)
    top = pick_bound(binding)
    roof = {"bound": top[0], "achieved": top[1]["achieved"], "peak": top[1]["peak"], "unit": top[1]["unit"],
        "frac": top[1]["frac"]} if top else \
           {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "note": "no calibration under profiles/"}
    compulsory = 48 * n + nodes.nbytes + tris.nbytes
    vi = (binding or {}).get("valu_issue") or {}
    hbm_alg = achieved / HBM_PEAK_GBPS
    per_s = 1.0 / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS          # bytes per launch -> fraction of the HBM peak
    stand_in = "" if top and top[0] == "valu_issue" else \
        "  [the committed counter pass does not belong to the running sources: the live node-fetch bound stands in]"
    top_r = pick_bound(binding_r)
    roof.update({
        "traffic": None if traffic is None else traffic["bytes"],
        "kernel": kname, "kernel_ms": round(k_mean, 5),
        "frac_of_measured_loop_mix_ceiling": vi.get("frac_of_measured_loop_mix_ceiling"), "lane_utilisation": vi.get("lane_utilisation"),
        # BASELINE's "fraction of HBM roofline", one key: fabric bytes of the committed --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE) / live
        # kernel time / 8 TB/s, beside the compulsory bytes (rays in, hits out, the BVH once) and the write amplification (WRITE_SIZE / the
        # 16-byte Hit1 array)
        "hbm": {"measured_frac": None if traffic is None else round(traffic["bytes"] * per_s, 4),
                "compulsory_frac": round(compulsory * per_s, 4),
                "traffic_over_compulsory": None if traffic is None else round(traffic["bytes"] / compulsory, 3),
                "write_amplification": None if traffic is None else round(traffic["write_bytes"] / (16.0 * n), 3),
                "peak_GBps": HBM_PEAK_GBPS, "source": None if traffic is None else traffic["source"]},
        "hbm_measured_frac": None if traffic is None else round(traffic["bytes"] * per_s, 4),
        "cache_served_bytes_over_hbm_peak": round(hbm_alg, 4),
        "cache_served_bytes_over_hbm_peak_is": "NOT a fraction: SURVEY 8(d)'s bytes per ray count every node / triangle visit although the "
            "BVH is "
                                               "served by LDS / L1 / L2 / MALL (> 1 is expected)",
        "what": "VALU issue: wave-instructions per us per SIMD of this kernel (SQ_INSTS_VALU of the committed counter pass / live kernel "
                "time / SIMDs) against the guide's 2-cycle rate at the measured clock.  At 1 Mi rays per launch the launch is a tail "
                "(LAB_NOTES.md 3.1.1); the same kernel at 16 Mi rays is in extra.primary_16Mi_rays_per_launch" + stand_in,
        "hbm_algorithmic": {"bound": "hbm", "GBps": round(achieved, 2), "peak_GBps": HBM_PEAK_GBPS, "frac": round(hbm_alg, 5),
                            "bytes_per_ray": round(bytes_per_ray, 2),
                            "visits_per_ray": {"inner": round(st["inner_per_ray"], 3), "prim": round(st["prims_per_ray"], 3)},
                            "note": "SURVEY 8(d): 32 + 16 + 64 N_inner + 48 N_tri bytes per ray x rays / kernel time: a count of cache "
                                "hits, not a "
                                    "fraction of anything"},
        # FETCH_SIZE x 2 = 128-byte line fills, calibrated on scattered 64-byte node fetches as well
        # (profiles/r04_fetch_size_calibration.txt): bytes between the L2s and the fabric, Infinity-Cache hits included -- an upper bound on
        # DRAM bytes
        "l2_fabric_traffic": {"not_quoted": traffic_why} if traffic is None else
                             {"what": "FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes: bytes between the L2s and the fabric, "
                                 "Infinity-Cache "
                                      "hits included (an upper bound on HBM bytes)",
                              "bytes_per_launch": traffic["bytes"], "GBps": round(traffic["bytes"] / (k_mean * 1e-3) / 1e9, 1),
                              "frac": round(traffic["bytes"] * per_s, 4), "compulsory_bytes_per_launch": int(compulsory),
                              "over_compulsory": round(traffic["bytes"] / compulsory, 3), "write_bytes": traffic["write_bytes"],
                              "write_over_hit1_array": round(traffic["write_bytes"] / (16.0 * n), 3), "source": traffic["source"]},
        "binding": binding,
        "random": {"kernel": kname, "kernel_ms": round(kr_mean, 5),
                   "visits_per_ray": {"inner": round(st_r["inner_per_ray"], 3), "prim": round(st_r["prims_per_ray"], 3)},
                   "bound": top_r[0] if top_r else None, "frac": top_r[1]["frac"] if top_r else None, "binding": binding_r}})
    # parity on every ray of both sets (bit-exact for the order-preserving kernels)
    parity = {"primary": bool(hits.tobytes() == ref_hits.tobytes()), "random": bool(hits_rnd.tobytes() == ref_rnd.tobytes())}
    return roof, parity
