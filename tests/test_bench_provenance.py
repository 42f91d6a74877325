"""bench.py quotes counter-derived numbers (VALU instructions, HBM bytes, per-kernel renderer figures) from profiles committed under
profiles/ -- only while the profile carries the hash of the kernel sources bench.py is running on and holds a pass of the very
kernel it timed (rodent_amd/provenance.py).  CPU-only: the selection logic, not the numbers."""
import importlib.util
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture()
def bench(tmp_path, monkeypatch):
    """benchlib/profiles.py (what bench.py reads committed profiles with), looking at an empty profiles/ directory."""
    sys.path.insert(0, str(ROOT))
    import benchlib.profiles as mod
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(mod, "ROOT", tmp_path)
    return mod


def test_source_hash_follows_the_sources():
    from rodent_amd import provenance
    a, b = provenance.source_sha("traversal"), provenance.source_sha("render")
    assert len(a) == 16 and a != b and provenance.is_current(provenance.stamp("traversal"), "traversal")
    assert not provenance.is_current(provenance.stamp("traversal"), "render") and not provenance.is_current(None,
        "traversal") and not provenance.is_current({}, "render")


def test_counters_are_quoted_only_from_a_profile_of_these_sources_and_this_kernel(bench, tmp_path):
    from rodent_amd import provenance
    kernel = "k_bvh2_top_persist<false,15, 255, 16, false>"
    groups = {"r09_pmc_primary_sq1": {"k_bvh2_top_persist<false, 15, 255, 16, false, 32, false, 0, 2, false, false>": {
        "SQ_INSTS_VALU": 7.0e7, "SQ_ACTIVE_INST_VALU": 7.1e7, "SQ_THREAD_CYCLES_VALU": 2.0e9}},
              "r09_pmc_random_sq1": {"k_other<1>": {"SQ_INSTS_VALU": 1.0}}}
    # no profile at all
    assert bench.kernel_counters(kernel, "primary")[0] == {} and "no profiles" in bench.kernel_counters(kernel, "primary")[1]
    # a profile of other sources: refused, with the reason
    (tmp_path / "profiles" / "r09_pmc_counters.json").write_text(json.dumps(dict(groups, _meta={"source_sha": "0123456789abcdef"})))
    c, why = bench.kernel_counters(kernel, "primary")
    assert c == {} and "other traversal sources" in why
    # a profile of these sources: the kernel's primary pass is quoted, the random pass (another kernel) is not
    (tmp_path / "profiles" / "r09_pmc_counters.json").write_text(json.dumps(dict(groups, _meta=provenance.stamp("traversal"))))
    c, why = bench.kernel_counters(kernel, "primary")
    assert why is None and c["SQ_INSTS_VALU"] == 7.0e7 and c["source"] == "r09_pmc_counters.json"
    c, why = bench.kernel_counters(kernel, "random")
    assert c == {} and "holds no pass of kernel" in why
    # the newest file decides (r10 is stale again)
    (tmp_path / "profiles" / "r10_pmc_counters.json").write_text(json.dumps(dict(groups, _meta={"source_sha": "ffffffffffffffff"})))
    assert bench.kernel_counters(kernel, "primary")[0] == {}
    # traffic: same rule
    (tmp_path / "profiles" / "r09_traffic.json").write_text(json.dumps({"_meta": provenance.stamp("traversal"),
        "k_bvh2_top_persist<false, 15, 255, 16, false, 32, false, 0, 2, false, false>": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0,
            "hbm_bytes_fetch_x2": 2560000.0}}))
    t, why = bench.measured_traffic(kernel)
    assert why is None and t["bytes"] == 2560000 and t["write_bytes"] == 512000
    (tmp_path / "profiles" / "r09_traffic.json").write_text(json.dumps({"_meta": provenance.stamp("render"), "x": {}}))
    assert bench.measured_traffic(kernel)[0] is None


def test_the_top_level_roofline_is_valu_issue_against_the_guides_rate(bench, tmp_path):
    """ONE roofline: VALU issue against the guide's 2-cycle rate at the measured clock; the measured loop-mix ceiling
    and the lane utilisation ride along; the round-2 figure 771 is gone from the calibration bench.py reads.  The live node-fetch bound
    stands in only while the committed counter pass does not belong to the running sources."""
    from rodent_amd import provenance
    b = {"vmem_node_fetch": {"frac": 0.80, "achieved": 1, "peak": 2, "unit": "fetches/ns"},
        "valu_issue": {"frac": 0.35, "achieved": 3, "peak": 4, "unit": "i"}, "lds_fetch": {"frac": 0.9}}
    assert bench.pick_bound(b)[0] == "valu_issue"
    del b["valu_issue"]                                            # counters not quoted: the live bound is all there is
    assert bench.pick_bound(b)[0] == "vmem_node_fetch" and bench.pick_bound({}) is None and bench.pick_bound(None) is None
    cal = json.loads(sorted((ROOT / "profiles").glob("r*_calibration.json"))[-1].read_text())
    assert "valu_issue_peak" not in cal and cal["valu_issue_guide_2_cycle_rate"] == 1162.0 and cal["valu_issue_peak_loop_mix_r04"] == 644.7
    (tmp_path / "profiles" / "r09_calibration.json").write_text(json.dumps(cal))
    kernel = "k_bvh2_top_auto<false,15, 255, 16, 32>"
    groups = {"r09_pmc_primary_sq1": {"k_bvh2_top_auto<false, 15, 255, 16, 32, 0, true>": {"SQ_INSTS_VALU": 7.7e7,
        "SQ_ACTIVE_INST_VALU": 7.8e7, "SQ_THREAD_CYCLES_VALU": 2.4e9}}, "_meta": provenance.stamp("traversal")}
    (tmp_path / "profiles" / "r09_pmc_counters.json").write_text(json.dumps(groups))
    vi = bench.binding_bounds(kernel, "primary", 1 << 20, 39.3, 0.19, 18.8)["valu_issue"]
    assert vi["peak"] == 1162.0 and abs(vi["achieved"] - 7.7e7 / 190.0 / 1024) < 0.1 and abs(vi["frac"] - vi["achieved"] / 1162.0) < 1e-3
    assert abs(vi["frac_of_measured_loop_mix_ceiling"] - vi["achieved"] / 644.7) < 1e-3 and abs(vi["lane_utilisation"] - 2.4e9 / 64
        / 7.8e7) < 1e-3


def test_renderer_profiles_are_checked_the_same_way(bench, tmp_path):
    from rodent_amd import provenance
    name = "cfg4_cornell_1920x1080_64spp_len4"
    assert "not_quoted" in bench.render_profile(name)
    prof = {"_meta": dict(provenance.stamp("render"), command="rodent ...", frames=2),
            "streaming": {"k_shade": {"calls_per_frame": 42.0, "avg_us": 340.0, "fetch_MB": 1.0, "write_MB": 2.0,
                "hbm_TBps_fetch_x2": 3.2}}}
    (tmp_path / "profiles" / "r09_render_profile_cfg4.json").write_text(json.dumps(prof))
    got = bench.render_profile(name)
    assert got["streaming"]["k_shade"] == {"calls_per_frame": 42.0, "avg_ms": 0.34, "ms_per_frame": 14.28, "hbm_frac": 0.4,
        "hbm_GBps": 3200.0}
    prof["_meta"]["source_sha"] = "0" * 16
    (tmp_path / "profiles" / "r09_render_profile_cfg4.json").write_text(json.dumps(prof))
    assert "other render sources" in bench.render_profile(name)["not_quoted"]
