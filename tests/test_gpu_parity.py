"""GPU parity tests: HIP kernels (through the C ABI) against the CPU oracle.

Bar (BASELINE.json): primitive ids bit-exact, t within 1e-4 relative.  Kernels that keep the
reference's per-ray visit order are held to the stronger bar: the whole Hit1 record
(id, t, u, v) bit-identical to the oracle, ties and any-hit results included.
"""
import ctypes
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN as GOLDEN_DIR, ambiguous_mask


def cornell_bvh_path():
    return GOLDEN_DIR / "cornell.bvh"

from rodent_amd import formats as F

pytestmark = pytest.mark.gpu

# (bvh width, golden/oracle algorithm name, oracle algo flag)
LAYOUTS = {2: ("bvh2_gpu", "ref"), 4: ("bvh4_gpu", "gpu"), 8: ("bvh8_gpu", "gpu")}
BLOCKS = {2: F.BVH2_TRI1, 4: F.BVH4_TRI4, 8: F.BVH8_TRI4}


@pytest.fixture(scope="module")
def gpu(native_build):
    import torch
    from rodent_amd import abi
    assert torch.cuda.is_available(), "these tests need a GPU"
    abi.lib().rodent_hip_phased_min_rays(0)       # the phased mappings suspend and resume rays at every launch size in this module
    abi.lib().rodent_hip_top_min_rays(0)          # ... and the default mapping takes its LDS-image kernel, not the small-launch kernel
    yield abi
    abi.lib().rodent_hip_phased_min_rays(-1)
    abi.lib().rodent_hip_top_min_rays(-1)


@pytest.fixture(scope="module")
def cornell_dev(gpu, cornell):
    return {w: gpu.DeviceBvh(w, *cornell.blocks[w], 0) for w in (2, 4, 8)}


@pytest.fixture(scope="module")
def atrium(gpu, oracle):
    from rodent_amd import scenes, raygen
    path = scenes.scene_bvh("atrium")
    blocks = {w: F.read_bvh(path, BLOCKS[w]) for w in (2, 4, 8)}
    eye, d, up, fov = scenes.CAMERAS["atrium"]
    n4, _ = F.read_bvh(path, F.BVH4_TRI4)
    lo, hi = raygen.scene_bounds(n4)

    class A:
        dev = {w: gpu.DeviceBvh(w, *blocks[w], 0) for w in (2, 4, 8)}
        host = blocks
        primary = raygen.primary_rays(eye, d, up, fov, 1024, 1024, 0.0, 5000.0)
        random = raygen.random_rays(lo, hi, 1 << 20, 42, 0.0, 1.0)
    return A


def variants(gpu, width):
    return gpu.order_preserving_variants(width)          # (the order-changing mapping "steal" has its own test)


@pytest.mark.parametrize("width", [2, 4, 8])
@pytest.mark.parametrize("rayset", ["primary", "primary_tmin", "random", "edge"])
def test_cornell_golden_bit_exact(gpu, cornell, cornell_dev, width, rayset):
    name, _ = LAYOUTS[width]
    rays = cornell.ray_sets[rayset]
    for any_hit in (False, True):
        exp = cornell.expected[f"{name}.{rayset}.{'any' if any_hit else 'closest'}"]
        for v in variants(gpu, width):
            got = gpu.traverse(cornell_dev[width], rays, any_hit=any_hit, variant=v)
            bad = np.nonzero(got.view("<u4").reshape(-1, 4) != exp.view("<u4").reshape(-1, 4))[0]
            assert len(bad) == 0, f"width {width} variant {v} any={any_hit}: first diff ray {bad[0]}: {got[bad[0]]} vs {exp[bad[0]]}"


@pytest.mark.parametrize("width", [2, 4, 8])
def test_reference_named_entry_points(gpu, cornell, cornell_dev, width):
    """amdgpu_{intersect,occluded}_single_ray1_bvh2_tri1 / hip_*_bvh8_tri4 (synchronous)."""
    name, _ = LAYOUTS[width]
    for any_hit in (False, True):
        got = gpu.traverse(cornell_dev[width], cornell.ray_sets["random"], any_hit=any_hit, variant=None)
        assert got.tobytes() == cornell.expected[f"{name}.random.{'any' if any_hit else 'closest'}"].tobytes()


@pytest.mark.parametrize("width", [2, 4, 8])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000])
def test_ragged_sizes(gpu, cornell, cornell_dev, width, n):
    name, _ = LAYOUTS[width]
    rays = cornell.ray_sets["primary"][:n]
    exp = cornell.expected[f"{name}.primary.closest"][:n]
    for v in variants(gpu, width):
        got = gpu.traverse(cornell_dev[width], rays, variant=v)
        assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("n", [8192 * 64 + 16384, 600_000, (1 << 20) + 12345, 3 << 20])
def test_chunk_mapping_is_a_bijection(gpu, cornell, cornell_dev, n):
    """The kernels map workgroups (k_bvh2_single) / tickets (k_bvh2_top_persist, the default) to ray chunks XCD-aware: every
    chunk must be traced exactly once at sizes with ragged tails and several dispatch rounds; the hit buffer starts as 0xFF."""
    import torch
    names = gpu.variants(2)
    base = cornell.ray_sets["primary"]
    rays = np.tile(base, (n + len(base) - 1) // len(base))[:n].copy()
    rays["org"][:, 0] += (np.arange(n, dtype=np.float32) % 977) * 1e-4          # not all copies identical
    rd = gpu.to_device(rays, 0)
    out = {}
    for name in ("top", "fast", "fast-noxcd"):
        hd = torch.full((n * 16,), 0xFF, dtype=torch.uint8, device="cuda:0")
        gpu.traverse_async(cornell_dev[2], rd, hd, n, False, names.index(name))
        torch.cuda.synchronize()
        out[name] = gpu.from_device(hd, F.HIT1)
    assert out["fast"].tobytes() == out["fast-noxcd"].tobytes() and out["top"].tobytes() == out["fast"].tobytes()
    assert (out["fast"]["tri_id"] >= -1).all() and (out["fast"]["tri_id"] < 64).all()
    for width in (4, 8):                                       # the wide kernels use the same mapping
        wn = gpu.variants(width)
        wide = {}
        # "top": tickets of the persistent form (k_wide_top_persist), as the BVH2 default
        for name in ("top", "single", "single-noxcd"):
            hd = torch.full((n * 16,), 0xFF, dtype=torch.uint8, device="cuda:0")
            gpu.traverse_async(cornell_dev[width], rd, hd, n, False, wn.index(name))
            torch.cuda.synchronize()
            wide[name] = gpu.from_device(hd, F.HIT1)
        assert wide["single"].tobytes() == wide["single-noxcd"].tobytes() == wide["top"].tobytes()
        assert np.array_equal(wide["single"]["tri_id"] >= 0, out["fast"]["tri_id"] >= 0)


def test_launches_on_several_streams_may_overlap(gpu, oracle):
    """Every (device, stream) has its own control words, deep-ray list, suspended-ray queues and sort buffers: four streams
    trace different ray sets of the deep-chain scene (every launch hands rays to its follow-up kernel) at the same time,
    repeatedly, with every shipped mapping (the phased one suspends and resumes, the sorted one permutes); all results exact."""
    import torch
    from conftest import chain_bvh2
    nodes, tris = chain_bvh2(40)
    bvh = gpu.DeviceBvh(2, nodes, tris, 0)
    rng = np.random.default_rng(11)
    sets = []
    for k in range(4):
        n = 20000 + 777 * k
        org = np.zeros((n, 3), "<f4"); org[:, :2] = rng.uniform(-4, 4, (n, 2)); org[k::3, 0] += 50.0
        rays = F.make_rays(org, np.tile(np.float32([0.001, 0.002, 1.0]), (n, 1)), 0.0, 1000.0)
        ref, _ = oracle.traverse(2, nodes, tris, rays)
        sets.append((rays, ref, gpu.to_device(rays, 0), torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0"), torch.cuda.Stream()))
    torch.cuda.synchronize()
    for rep in range(5):
        for v in variants(gpu, 2):
            for rays, ref, rd, hd, st in sets:
                hd.fill_(0xFF)
            torch.cuda.synchronize()
            for k, (rays, ref, rd, hd, st) in enumerate(sets):
                # same / mixed mappings in flight
                gpu.traverse_async(bvh, rd, hd, len(rays), False, (v + k) % len(variants(gpu, 2)) if rep % 2 else v, st)
            torch.cuda.synchronize()
            for rays, ref, rd, hd, st in sets:
                assert gpu.from_device(hd, F.HIT1).tobytes() == ref.tobytes(), (rep, gpu.variants(2)[v])
        for st in (s[4] for s in sets):
            assert gpu.lib().rodent_hip_check_errors(0, ctypes.c_void_p(st.cuda_stream)) == 0


def test_special_tmin_tmax_values(gpu, oracle, cornell, cornell_dev):
    """tmin / tmax taken from {0, -0, denormal, +-inf, quiet NaN, signalling NaN, ...}: the reference's fminf / fmaxf box test
    ignores a NaN bound, the triangle test's comparisons reject it; every variant must reproduce the oracle bit for bit,
    the NaN payload of a miss record included (the default kernel issues raw v_max/v_min on canonicalised bounds)."""
    specials = np.array([0x00000000, 0x80000000, 0x00000001, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7FA00001, 0xFFC12345,
                         0x40A00000, 0xBF800000, 0x3C23D70A, 0x7F7FFFFF], dtype="<u4").view("<f4")
    base = cornell.ray_sets["primary"][:len(specials) ** 2 * 4].copy()
    k = np.arange(len(base))
    base["tmin"] = specials[k % len(specials)]
    base["tmax"] = specials[(k // len(specials)) % len(specials)]
    nodes, tris = F.read_bvh(cornell.bvh_path, F.BVH2_TRI1)
    for any_hit in (False, True):
        exp, _ = oracle.traverse(2, nodes, tris, base, any_hit=any_hit)
        for v in variants(gpu, 2):
            got = gpu.traverse(cornell_dev[2], base, any_hit=any_hit, variant=v)
            if any_hit:
                assert np.array_equal(got["tri_id"] >= 0, exp["tri_id"] >= 0), gpu.variants(2)[v]
            else:
                assert got.tobytes() == exp.tobytes(), gpu.variants(2)[v]


def test_deep_stack_falls_back_to_global_stack(gpu, oracle):
    """Stacks deeper than the LDS window (15 / 16 entries) move their oldest entries to the wave's block of global memory
    (stack_spill; the wide kernels and the launches without a spill block: deep-ray list + follow-up kernel);
    results must not change.  Mix deep and shallow rays in one wave and across waves."""
    from conftest import chain_bvh2
    nodes, tris = chain_bvh2(40)
    rng = np.random.default_rng(0)
    org = np.zeros((1000, 3), "<f4"); org[:, :2] = rng.uniform(-4, 4, (1000, 2)); org[::3, 0] += 50.0   # every 3rd ray misses
    d = np.tile(np.float32([0.001, 0.002, 1.0]), (1000, 1))
    rays = F.make_rays(org, d, 0.0, 1000.0)
    ref, st = oracle.traverse(2, nodes, tris, rays)
    assert st["max_stack"] == 40 and (ref["tri_id"] == 0).sum() > 600
    bvh = gpu.DeviceBvh(2, nodes, tris, 0)
    for any_hit in (False, True):
        ref, _ = oracle.traverse(2, nodes, tris, rays, any_hit=any_hit)
        for v in variants(gpu, 2):
            got = gpu.traverse(bvh, rays, any_hit=any_hit, variant=v)
            assert got.tobytes() == ref.tobytes(), f"variant {v} any={any_hit}"
        # and again: the launch counters must have been reset by the epilogue
        assert gpu.traverse(bvh, rays, any_hit=any_hit, variant=0).tobytes() == ref.tobytes()
    # stats word 7 = blocks of stack entries moved out (+ rays handed to a follow-up kernel): only rays that enter the 40-deep chain
    # spill, four times each (at 15, 22, 29 and 36 entries); a launch on a shallow tree touches nothing
    gpu.read_stats()
    gpu.traverse(bvh, rays, variant=0)
    deep = gpu.read_stats()[7]
    assert 0 < deep <= 4 * int((ref["tri_id"] >= 0).sum())
    cb = gpu.DeviceBvh.load(cornell_bvh_path(), 2, 0)
    gpu.traverse(cb, F.read_rays(GOLDEN_DIR / "cornell-primary-64x64.rays", 0.0, 100.0), variant=0)
    assert gpu.read_stats()[7] == 0


@pytest.mark.parametrize("depth", [16, 23, 40, 63])
def test_deep_stacks_in_the_persistent_kernels(gpu, oracle, depth):
    """The same through the persistent kernels (rodent_hip_top_min_rays(0): the default mapping's k_bvh2_top_auto / k_bvh2_top_refill,
    "refill", and the wide kernels' persistent form), at depths around every block boundary of the spill (15 + 7 k entries) up to the
    reference's capacity of 63 entries (stack.impala:53: 64 slots, one of them the sentinel), 30 000 rays of which a third miss, coherent
    and shuffled (the refill loop), twice (the second launch runs on the image the first one built), closest and any hit."""
    from conftest import chain_bvh2
    nodes, tris = chain_bvh2(depth)
    rng = np.random.default_rng(depth)
    n = 30000
    org = np.zeros((n, 3), "<f4"); org[:, :2] = rng.uniform(-4, 4, (n, 2)); org[::3, 0] += 50.0
    # no common direction: the refill loop
    d = np.tile(np.float32([0.001, 0.002, 1.0]), (n, 1)); d[:, :2] += rng.uniform(-1e-4, 1e-4, (n, 2)).astype("<f4")
    rays = F.make_rays(org, d, 0.0, 1000.0)
    bvh = gpu.DeviceBvh(2, nodes, tris, 0)
    gpu.lib().rodent_hip_top_min_rays(0)
    try:
        for any_hit in (False, True):
            ref, st = oracle.traverse(2, nodes, tris, rays, any_hit=any_hit)
            if not any_hit:
                assert st["max_stack"] == depth
            for v in variants(gpu, 2):
                for rep in range(2):
                    assert gpu.traverse(bvh, rays, any_hit=any_hit, variant=v).tobytes() == ref.tobytes(), (gpu.variants(2)[v], any_hit,
                        rep)
        gpu.read_stats()
        gpu.traverse(bvh, rays, variant=0)
        assert gpu.read_stats()[7] > 0
    finally:
        gpu.lib().rodent_hip_top_min_rays(-1)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_triangle_soups_bit_exact(gpu, oracle, native_build, tmp_path, seed):
    """Unstructured input through the whole tool chain (OBJ -> bvh_extractor -> .bvh -> HIP): a soup of random triangles with
    zero-area, needle, duplicated, shared-edge and huge-coordinate members; random, axis-parallel, grazing and
    vertex-aimed rays; every BVH2 variant and the BVH8 kernel bit for bit against the oracle, closest and any hit."""
    rng = np.random.default_rng(seed)
    nt = 700
    c = rng.uniform(-10, 10, (nt, 1, 3)); tri = c + rng.normal(0, [0.3, 1.5, 0.05][seed - 1], (nt, 3, 3))
    tri[:20, 2] = tri[:20, 1]                                   # zero area (two equal vertices)
    tri[20:40, 2] = tri[20:40, 0] + 1e3 * (tri[20:40, 1] - tri[20:40, 0])       # needles
    tri[40:60] = tri[60:80]                                     # exact duplicates (ties)
    tri[80:100, 0] = tri[100:120, 0]; tri[80:100, 1] = tri[100:120, 1]            # shared edges
    tri[120:125] *= 1e4                                         # huge coordinates
    tri = tri.astype(np.float32)
    with open(tmp_path / "soup.obj", "w") as f:
        for v in tri.reshape(-1, 3):
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for k in range(nt):
            f.write("f %d %d %d\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))
    subprocess.run([native_build.BIN_DIR / "bvh_extractor", "-obj", tmp_path / "soup.obj", "-o", tmp_path / "soup.bvh"], check=True,
        stdout=subprocess.DEVNULL)
    n = 6000
    org = rng.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[:500, rng.integers(0, 3, 500)] *= 1.0                     # (general rays)
    # axis-parallel: the reference's slab test culls them
    d[500:800] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1, 1], (300, 1))
    aim = tri.reshape(-1, 3)[rng.integers(0, 3 * nt, 1500)]
    d[800:2300] = aim - org[800:2300]                           # aimed at vertices (ties between neighbours)
    mid = 0.5 * (tri[rng.integers(0, nt, 700), 0] + tri[rng.integers(0, nt, 700), 1])
    d[2300:3000] = mid - org[2300:3000]
    rays = F.make_rays(org, d, 0.0, 3.0)                        # dir is not normalised: t in units of |dir|
    rays["tmin"][3000:3500] = 0.5; rays["tmax"][3500:4000] = 0.7
    for width in (2, 4, 8):
        name, algo = LAYOUTS[width]
        nodes, prims = F.read_bvh(tmp_path / "soup.bvh", BLOCKS[width])
        bvh = gpu.DeviceBvh(width, nodes, prims, 0)
        for any_hit in (False, True):
            ref, _ = oracle.traverse(width, nodes, prims, rays, any_hit=any_hit, algo=algo)
            assert (ref["tri_id"] >= 0).sum() > 100
            for v in variants(gpu, width):
                got = gpu.traverse(bvh, rays, any_hit=any_hit, variant=v)
                assert got.tobytes() == ref.tobytes(), f"BVH{width} variant {gpu.variants(width)[v]} any={any_hit}"


@pytest.mark.parametrize("width", [2, 4, 8])
@pytest.mark.parametrize("kind", ["primary", "random"])
def test_atrium_sample_bit_exact_vs_oracle(gpu, oracle, atrium, width, kind):
    """64 Ki rays of the benchmark dumps (every 16th ray) against the oracle run live."""
    _, algo = LAYOUTS[width]
    rays = getattr(atrium, kind)[::16]
    for any_hit in (False, True):
        ref, st = oracle.traverse(width, *atrium.host[width], rays, any_hit=any_hit, algo=algo)
        assert st["max_stack"] < 64
        for v in variants(gpu, width):
            got = gpu.traverse(atrium.dev[width], rays, any_hit=any_hit, variant=v)
            assert np.array_equal(got["tri_id"], ref["tri_id"]), f"ids differ (variant {v})"
            assert got.tobytes() == ref.tobytes(), f"t/u/v bits differ (variant {v})"


@pytest.mark.parametrize("kind", ["primary", "random"])
def test_atrium_cross_layout_parity(gpu, oracle, atrium, kind):
    """North-star bar against the reference's CPU traversal (oracle B2 on the BVH8 block):
    ids exact except on order-dependent ties, t within 1e-4 relative -- for both GPU layouts."""
    rays = getattr(atrium, kind)[::64]
    cpu, _ = oracle.traverse(8, *atrium.host[8], rays, algo="ref")
    for width in (2, 4, 8):
        got = gpu.traverse(atrium.dev[width], rays, variant=0)
        assert np.array_equal(got["tri_id"] >= 0, cpu["tri_id"] >= 0)
        hit = cpu["tri_id"] >= 0
        assert np.allclose(got["t"][hit], cpu["t"][hit], rtol=1e-4, atol=0)
        differ = got["tri_id"] != cpu["tri_id"]
        # every id difference must be a genuine tie: the two primitives are hit at (almost) the same t
        assert differ.mean() < 0.02
        if differ.any():
            sub = rays[differ]
            b2, s2 = oracle.brute_force(atrium.host[2][1], sub) if len(sub) <= 4096 else (None, None)
            if b2 is not None:
                assert ambiguous_mask(b2, s2).all()


@pytest.mark.parametrize("kind", ["primary", "random"])
def test_full_size_properties(gpu, atrium, kind):
    """1 Mi rays (BASELINE size): properties that need no oracle run."""
    rays = getattr(atrium, kind)
    base = gpu.traverse(atrium.dev[2], rays, variant=0)
    # all BVH2 mappings keep the per-ray order => identical bits
    for v in variants(gpu, 2)[1:]:
        assert gpu.traverse(atrium.dev[2], rays, variant=v).tobytes() == base.tobytes()
    hit = base["tri_id"] >= 0
    # misses return tmax untouched, hits lie inside [tmin, tmax]
    assert np.array_equal(base["t"][~hit], rays["tmax"][~hit])
    assert (base["t"][hit] >= rays["tmin"][hit]).all() and (base["t"][hit] <= rays["tmax"][hit]).all()
    assert (base["u"][hit] >= 0).all() and (base["v"][hit] >= 0).all() and (base["u"][hit] + base["v"][hit] <= 1 + 1e-5).all()
    # any-hit agrees with closest-hit on occlusion
    occ = gpu.traverse(atrium.dev[2], rays, any_hit=True, variant=0)
    assert np.array_equal(occ["tri_id"] >= 0, hit)
    # the BVH4 / BVH8 layouts agree on hit/miss and on t to 1e-4
    for width in (4, 8):
        wide = gpu.traverse(atrium.dev[width], rays, variant=0)
        assert np.array_equal(wide["tri_id"] >= 0, hit)
        assert np.allclose(wide["t"][hit], base["t"][hit], rtol=1e-4, atol=0)
        assert (wide["tri_id"] != base["tri_id"]).mean() < 0.02
        for v in variants(gpu, width)[1:]:
            assert gpu.traverse(atrium.dev[width], rays, variant=v).tobytes() == wide.tobytes()
    # idempotence: shrinking tmax to just beyond the hit returns the same primitive
    sub = rays[::8].copy()
    sub["tmax"] = np.where(hit[::8], base["t"][::8] * np.float32(1.0001), sub["tmax"])
    again = gpu.traverse(atrium.dev[2], sub, variant=0)
    assert (again["tri_id"] != base["tri_id"][::8]).mean() < 2e-3      # only near-ties may flip (different culling history)
    # (rtol 2e-4: with tmax this tight the reference's slab test -- inv_org = -(org * inv_dir), absolute error
    # ~6e-8 * |org * inv_dir| -- may cull the box that holds the hit, and the ray then reports tmax = t * 1.0001)
    assert np.allclose(again["t"], np.where(hit[::8], base["t"][::8], sub["tmax"]), rtol=2e-4, atol=0)
    if kind == "primary":
        assert hit.mean() > 0.9999                        # closed scene seen from inside (a few rays slip through cracks)
    else:
        assert 0.05 < hit.mean() < 0.95


def test_bench_traversal_cli(gpu, native_build, oracle, cornell, tmp_path):
    """The CLI keeps the reference's stdout protocol (bench_traversal.cpp:294,381-391) and .fbuf."""
    out = tmp_path / "o.fbuf"
    cmd = [native_build.BIN_DIR / "bench_traversal", "-bvh", cornell.bvh_path, "-ray",
        cornell.bvh_path.parent / "cornell-primary-64x64.rays",
           "--tmin", "0.01", "--tmax", "5000", "--bench", "3", "--warmup", "1", "-o", out]
    for extra, name in ((["-gpu", "amdgpu"], "bvh2_gpu"), (["-gpu", "hip", "--variant", "1"], "bvh2_gpu"),
        (["-gpu", "hip", "--bvh-width", "4"], "bvh4_gpu"),
                        (["-gpu", "hip", "--bvh-width", "8"], "bvh8_gpu")):
        r = subprocess.run(cmd + extra, capture_output=True, text=True, check=True)
        lines = r.stdout.strip().splitlines()
        assert lines[0] == "4096 ray(s) in the distribution file."
        assert lines[1].endswith("ms for 3 iteration(s)") and lines[2].endswith(" Mrays/sec")
        assert lines[3].startswith("# Average: ") and lines[4].startswith("# Median: ") and lines[5].startswith("# Min: ")
        assert lines[6] == "4096 intersection(s)"
        exp = cornell.expected[f"{name}.primary_tmin.closest"]
        assert np.array_equal(F.read_fbuf(out), exp["t"])
    r = subprocess.run(cmd + ["-gpu", "amdgpu", "-any"], capture_output=True, text=True, check=True)
    assert "4096 intersection(s)" in r.stdout
    # -ngpu K: contiguous ray ranges, one Hit1 gather to the first device; K = 1 and every GPU the box has give the same file
    have = gpu.lib().rodent_hip_device_count()
    for k in sorted({1, min(have, 2), have}):
        r = subprocess.run(cmd + ["-gpu", "hip", "-ngpu", str(k), "--hits", tmp_path / "h.bin"], capture_output=True, text=True, check=True)
        assert "4096 intersection(s)" in r.stdout and (k == 1 or f"# GPUs: {k}" in r.stdout)
        assert np.array_equal(F.read_fbuf(out), cornell.expected["bvh2_gpu.primary_tmin.closest"]["t"])
        hits = np.fromfile(tmp_path / "h.bin", F.HIT1)
        assert np.array_equal(hits["t"], cornell.expected["bvh2_gpu.primary_tmin.closest"]["t"]) and (hits["tri_id"] >= 0).all()
    r = subprocess.run(cmd + ["-gpu", "hip", "-ngpu", str(have + 1)], capture_output=True, text=True)
    assert r.returncode != 0 and "No such GPU device(s)" in r.stderr
    # The K > 1 path on however few GPUs the box has: RODENT_SHARE_GPUS=1 puts the K ranks' threads, ray ranges and Hit1
    # pieces on the devices there are; RCCL cannot place two ranks on one device, so the gather takes its fallback (one hipMemcpyPeerAsync
    # per piece) and says so; the tool then checks the assembled array against one device's trace of all rays by itself.  An injected RCCL
    # failure takes the same branch.
    import os
    for k, env in ((3, {"RODENT_SHARE_GPUS": "1"}), (min(have, 2), {"RODENT_FORCE_RCCL_INIT_FAILURE": "1"})):
        if k < 2:
            continue
        r = subprocess.run(cmd + ["-gpu", "hip", "-ngpu", str(k), "--hits", tmp_path / "h.bin"], capture_output=True, text=True,
            env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert f"# GPUs: {k}" in r.stdout and "# Collective: hipMemcpyPeerAsync per piece" in r.stdout
        assert "WARNING: RCCL is not used" in r.stderr
        assert "# Check: the gathered Hit1 array EQUALS" in r.stdout and "# Kernel ms per rank" in r.stdout
        hits = np.fromfile(tmp_path / "h.bin", F.HIT1)
        assert np.array_equal(hits["t"], cornell.expected["bvh2_gpu.primary_tmin.closest"]["t"])


def test_bench_py_contract(native_build):
    """bench.py prints ONE JSON line with the contract's keys, the roofline and cpu_baseline objects, the binding bounds, and
    reports every ray of both sets bit-exact against the oracle (variant 0 = what the driver runs)."""
    import json, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
        check=True, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d[
        "vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    # ONE top-level roofline: VALU issue against the guide's 2-cycle rate (the live node-fetch bound stands in while the committed counter
    # pass is stale): a fraction never above 1; the measured and the algorithmic HBM fractions side by side at the top level, the latter
    # labelled as a count of cache hits
    assert rf["bound"] in ("valu_issue",
        "vmem_node_fetch") and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3
    assert rf["bound"] != "valu_issue" or (rf["peak"] == 1162.0 and 0 < rf["frac_of_measured_loop_mix_ceiling"] <= 1.0
        and 0 < rf["lane_utilisation"] <= 1.0)
    # BASELINE's "fraction of HBM roofline" is ONE key: roofline.hbm (measured fabric bytes of the committed --pmc passes / kernel time / 8
    # TB/s); the survey's bytes-per-visit figure is a count of cache hits and is named so (it is no fraction: > 1 on a cache-resident tree)
    assert rf["cache_served_bytes_over_hbm_peak"] > 0 and "hbm_algorithmic_frac" not in rf and "compulsory_frac" in rf["hbm"]
    assert rf["traffic"] is None or (0 < rf["hbm"]["measured_frac"] < 1.0 and rf["hbm"]["traffic_over_compulsory"] > 0.9
        and rf["hbm"]["write_amplification"] > 0.9)
    assert "random_Mrays_s" in d["config"] and "random_with_kind_hint_Mrays_s" in d["config"]
    # the other scene classes and the any-hit ray class ride along: every cell's sample checked against the oracle
    # inside bench.py
    assert d["config"]["scene_classes_parity"] is True and d["config"]["ao_Mrays_s"] > 500 and set(d["extra"]["scenes"]) >= {"gallery",
        "crown", "plant"}
    assert all(d["extra"]["scenes"][k][c]["beyond_window_share"] < 0.5 for k in ("gallery", "crown", "plant") for c in ("primary",
        "random", "ao"))
    assert rf["hbm_algorithmic"]["bound"] == "hbm" and rf["hbm_algorithmic"]["peak_GBps"] == 8000.0 and rf["hbm_algorithmic"][
        "bytes_per_ray"] > 48
    assert 0 < rf["binding"]["vmem_node_fetch"]["frac"] < 1.0 and 0 < rf["random"]["binding"]["vmem_node_fetch"]["frac"] < 1.2
    # counter-derived figures are quoted only from a profile of THESE kernel sources (rodent_amd/provenance.py)
    from rodent_amd import provenance
    import glob
    pmc = sorted(glob.glob(str(ROOT / "profiles" / "r*_pmc_counters.json")))
    current = bool(pmc) and provenance.is_current(json.load(open(pmc[-1])).get("_meta"), "traversal")
    assert ("valu_issue" in rf["binding"]) == current and ("counters_not_quoted" in rf["binding"]) == (not current)
    assert not current or 1.0 < rf["binding"]["occupancy"]["resident_waves_per_simd_time_averaged_profiled"] <= 8.0
    rd = d["extra"]["render"]
    for cfg, (w, h, spp) in (("cfg4_cornell_1920x1080_64spp_len4", (1920, 1080, 64)),
        ("cfg5_atrium_3840x2160_256spp_len8", (3840, 2160, 256))):
        e = rd[cfg]
        assert (e["width"], e["height"], e["spp"]) == (w, h,
            spp) and e["auto"]["Msamples_s"] > 100 and e["auto"]["rays"]["generated"] == w * h * spp
        assert e["auto_mapping"] in ("streaming",
            "megakernel") and all(m in e for m in ("streaming", "streaming_sorted",
            "megakernel")) and e["streaming_sorted"]["Msamples_s"] > 100
    assert rd["cpu_baseline"]["cfg4_cornell_1920x1080_64spp_len4"]["Msamples_s"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["passes"] >= 10 and "sample" in cb
    assert d["value"] > 1000 and d["extra"]["all_rays_bit_exact_vs_oracle"] == {"primary": True, "random": True}
    assert d["extra"]["library"]["built_from_these_sources"] is True
    assert d["extra"]["random_sorted"]["identical_to_unsorted"] is True
    assert d["extra"]["random_refill"]["identical_to_default"] is True
    assert d["extra"]["random_8Mi_rays_per_launch"]["identical_hits"] is True


def test_bench_py_with_two_ranks(native_build):
    """The N > 1 code path of bench.py -- strong partition as `value` (BASELINE's metric: ONE 1 Mi-ray set, contiguous ranges, Hit1 gather
    to rank 0 after the timed region, assembled array equal to a single-GPU trace), weak partition beside it (1 Mi rays per GPU), config 5
    as interleaved 16-row tiles with a film gather -- with two ranks.  On a box with one GPU the ranks share it and the collectives go
    through gloo (RODENT_BENCH_SHARE_GPUS / _BACKEND); with two or more GPUs this is the driver's RCCL launch."""
    import json, os, sys
    import torch
    from conftest import ROOT
    env = dict(os.environ, MASTER_PORT="29541")
    if torch.cuda.device_count() < 2:
        env.update(RODENT_BENCH_BACKEND="gloo", RODENT_BENCH_SHARE_GPUS="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--render-spp5", "2"],
        capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rays_per_gpu_per_step"] == (1 << 19) and d[
        "value"] > 1000 and "per GPU" not in d["config"]["workload"]
    e = d["extra"]
    assert e["strong_scaling_check"] == {"primary_equal_to_single_gpu": True, "random_equal_to_single_gpu": True}
    assert e["weak_scaling"]["rays_per_gpu_per_step"] == (1 << 20) and e["weak_scaling"]["Mrays_s"] > 1000 and len(
        e["weak_scaling"]["kernel_ms_per_rank[primary,random]"]) == 2
    assert e["strong_scaling"]["rays_per_gpu_per_step"] == (1 << 19) and e["strong_scaling"]["Mrays_s"] == d["value"]
    assert d["config"]["strong_scaling_Mrays_s[primary,random]"][0] == e["strong_scaling"]["Mrays_s"] and "predicted_scaling_x" in d[
        "config"]
    assert e["all_rays_bit_exact_vs_oracle"] == {"primary": True, "random": True}          # rank 0's range against the oracle
    c5 = e["render"]["cfg5_atrium_3840x2160_256spp_len8"]
    # rank 0: tiles 0, 2, ..., 134 of 135
    assert c5["rows_per_gpu"] == 68 * 16 and "interleaved" in c5["partition"] and c5["auto"]["film_complete_on_root"] is True
    assert c5["auto"]["film_gather_ms"] > 0 and "cfg4_cornell_1920x1080_64spp_len4" not in e["render"]
    assert "cpu_baseline" not in d                                                          # rank 0, N = 1 only


def test_stack_overflow_is_reported_not_silent(gpu, oracle):
    """The reference's stack holds 64 entries, unchecked (stack.impala:53-54).  A ray that needs more raises a device flag:
    rodent_hip_check_errors reports it once and clears it (the synchronous entry points abort on it); launches that stay
    within 64 entries are unaffected afterwards."""
    from conftest import chain_bvh2
    nodes, tris = chain_bvh2(70)                                 # 70 entries deep
    org = np.zeros((200, 3), "<f4"); org[:, :2] = np.random.default_rng(2).uniform(-4, 4, (200, 2))
    rays = F.make_rays(org, np.tile(np.float32([0.001, 0.002, 1.0]), (200, 1)), 0.0, 1000.0)
    deep = gpu.DeviceBvh(2, nodes, tris, 0)
    for v in variants(gpu, 2):
        with pytest.raises(RuntimeError, match="stack overflow"):
            gpu.traverse(deep, rays, variant=v)
        assert gpu.lib().rodent_hip_check_errors(0, None) == 0   # cleared by the report
    ok_nodes, ok_tris = chain_bvh2(60)
    ok = gpu.DeviceBvh(2, ok_nodes, ok_tris, 0)
    ref, st = oracle.traverse(2, ok_nodes, ok_tris, rays)
    assert st["max_stack"] == 60
    assert gpu.traverse(ok, rays, variant=0).tobytes() == ref.tobytes()



def test_schedule_history_only_reorders_the_chunks(gpu, oracle, cornell, cornell_dev):
    """rodent_hip_schedule_history(1): a launch traces its chunks in the order the previous launch of the same size sorted them
    into (longest first).  Whatever the order -- none yet, one from the same rays, one from OTHER rays of the same count, ragged
    sizes -- every ray must be traced exactly once and get the oracle's hit."""
    import torch
    top = gpu.variants(2).index("top")
    base = cornell.ray_sets["primary"]
    nodes, tris = cornell.blocks[2]
    st = torch.cuda.Stream()
    gpu.lib().rodent_hip_schedule_history(1)
    try:
        # (100 rays and 34 x 64 rays: stripes of exactly two chunks -- the follow-up kernel's rank count reads its keys four at a time)
        for k, n in enumerate((100_000, 100_000, 100_000, 70_001, 70_001, 100_000, 4096 * 64 + 7, 4096 * 64 + 7, 100, 100, 100, 34 * 64,
            34 * 64, 34 * 64)):
            rays = np.tile(base, (n + len(base) - 1) // len(base))[:n].copy()
            rays["org"][:, 0] += (np.arange(n, dtype=np.float32) % 977) * 1e-4
            if k == 2:
                rays = rays[::-1].copy()                       # same count, other rays: the history mispredicts, nothing else
            rd = gpu.to_device(rays, 0)
            hd = torch.full((n * 16,), 0xFF, dtype=torch.uint8, device="cuda:0")
            # (the fill runs on torch's stream, the launch on `st`: without this the fill can land on top of the hits)
            torch.cuda.synchronize()
            gpu.traverse_async(cornell_dev[2], rd, hd, n, False, top, st)
            gpu.check_errors(0, st)
            ref, _ = oracle.traverse(2, nodes, tris, rays)
            assert gpu.from_device(hd, F.HIT1).tobytes() == ref.tobytes(), (k, n)
    finally:
        gpu.lib().rodent_hip_schedule_history(0)


def test_default_mapping_switches_kernels_at_its_size_threshold(gpu, oracle, cornell, cornell_dev):
    """With the shipped threshold (rodent_hip_top_min_rays(-1): 393 216 rays) launches just under it take the one-chunk kernel and
    launches from it on the persistent LDS-image kernel (stats[6]: workgroups that ran on the image); the hits are the oracle's."""
    import torch
    top = gpu.variants(2).index("top")
    base = cornell.ray_sets["primary"]
    nodes, tris = cornell.blocks[2]
    gpu.lib().rodent_hip_top_min_rays(-1)
    try:
        for n, persistent in ((6144 * 64 - 1, False), (6144 * 64, True), (6144 * 64 + 1, True)):
            rays = np.tile(base, (n + len(base) - 1) // len(base))[:n].copy()
            rays["org"][:, 1] += (np.arange(n, dtype=np.float32) % 613) * 1e-4
            rd = gpu.to_device(rays, 0)
            hd = torch.full((n * 16,), 0xFF, dtype=torch.uint8, device="cuda:0")
            for _ in range(2):                                  # the second launch finds the image the first one left
                gpu.read_stats(0)
                gpu.traverse_async(cornell_dev[2], rd, hd, n, False, top)
                gpu.check_errors(0)
            assert (gpu.read_stats(0)[6] > 0) == persistent, n
            ref, _ = oracle.traverse(2, nodes, tris, rays)
            assert gpu.from_device(hd, F.HIT1).tobytes() == ref.tobytes(), n
    finally:
        gpu.lib().rodent_hip_top_min_rays(0)


@pytest.mark.gpu
def test_camera_rays_in_image_order_are_traced_as_tiles(gpu, oracle, cornell, cornell_dev):
    """Round 5: the default BVH2 kernel recognises the pixels of an image, row by row (the reference's primary-ray dumps,
    tools/ray_gen/ray_gen.cpp:20-58), from 66 of the rays and gives every wavefront an 8 x 8-pixel tile instead of 64 pixels of a row
    (detect_ray_grid, traversal_top.h; stats[2] = the width it used).  Whatever it recognises -- widths that are no multiple of 8, heights
    that are not (the last rows stay in list order), two images in one list, normalised directions, segments in no order, a width forced on
    rays that are no image at all -- every Hit1 record is the oracle's, in its ray's place."""
    import torch
    from rodent_amd import raygen, scenes
    top = gpu.variants(2).index("top")
    nodes, tris = cornell.blocks[2]
    cam = scenes.CAMERAS["cornell"]
    lo, hi = raygen.scene_bounds2(nodes)

    def run(rays, expect_width, force=-1):
        gpu.ray_grid(force)
        try:
            gpu.read_stats(0)
            got = gpu.traverse(cornell_dev[2], rays, variant=top)
            gpu.check_errors(0)
            used = int(gpu.read_stats(0)[2])
        finally:
            gpu.ray_grid(-1)
        ref, _ = oracle.traverse(2, nodes, tris, rays)
        assert got.tobytes() == ref.tobytes(), (len(rays), expect_width, force)
        if expect_width is not None:
            assert used == expect_width, (len(rays), used, expect_width, force)
        for wide in (4, 8):                                              # the BVH4 / BVH8 kernels map their chunks the same way
            gpu.ray_grid(force)
            try:
                got = gpu.traverse(cornell_dev[wide], rays, variant=0)
            finally:
                gpu.ray_grid(-1)
            ref, _ = oracle.traverse(wide, *cornell.blocks[wide], rays, algo="gpu")
            assert got.tobytes() == ref.tobytes(), (wide, len(rays), expect_width, force)

    for w, h, expect in ((256, 64, 256), (1024, 24, 1024), (136, 50, 136), (1000, 16, 1000), (1004, 12, 0), (128, 40, 128), (120, 40, 0),
        (8192, 8, 8192), (8200, 8, 0), (1920, 33, 1920)):
        run(raygen.primary_rays(*cam, w, h, 0.0, 5000.0), expect)
    image = raygen.primary_rays(*cam, 256, 64, 0.0, 5000.0)
    run(image, 0, force=0)                                               # switched off: list order
    # a second image behind the first (here: the same pixels backwards) is traced by the first one's tiles
    run(np.concatenate([image, image[::-1]]), 256)
    unit = image.copy()
    unit["dir"] /= np.linalg.norm(unit["dir"], axis=1, keepdims=True)
    # normalised directions are no ray_gen dump: recognised or not, the hits are right
    run(unit, None)
    segments = raygen.random_rays(lo, hi, 40_000, 7, 0.0, 1.0)
    run(segments, 0)
    # An image AND segments in one list: the width is recognised, the image's waves trace tiles, the segments' waves refill
    # lanes -- and both must map the launch's positions onto its rays the same way.  Heights that are no multiple of 8: the image's last
    # rows share a band of 8 rows with segments. (136 x 53 = 7 208 rays: the probes beyond them are segments -- recognised or not)
    for w, h, expect in ((256, 60, 256), (1024, 20, 1024), (136, 53, None)):
        run(np.concatenate([raygen.primary_rays(*cam, w, h, 0.0, 5000.0), segments]), expect)
    run(np.concatenate([segments[:1000], image]), None)               # (segments first: whatever is recognised, the hits are right)
    # per-pixel lists that are no camera dump (the width is read from the pixel BELOW ray 0 being a near neighbour: multiples of 128 that
    # divide the ray count)
    light = np.array([0.0, 1.9, 0.0], np.float32)
    for w, h, expect in ((256, 64, 256), (1024, 24, 1024), (384, 40, 384), (200, 64, 0)):
        cam_rays = raygen.primary_rays(*cam, w, h, 0.0, 5000.0)
        t = oracle.traverse(2, nodes, tris, cam_rays)[0]["t"]
        # ray_gen's shadow mode: from a light to the hit points (the suite's "ao" class)
        run(raygen.shadow_rays(light, cam_rays, t, 0.0, 0.999), expect)
        hit_points = cam_rays["org"] + t[:, None] * cam_rays["dir"]
        # a renderer's: from the hit points to the light
        run(F.make_rays(hit_points.astype(np.float32), (light - hit_points).astype(np.float32), 0.001, 0.999), expect)
        # a count the width does not divide: list order
        run(raygen.shadow_rays(light, cam_rays, t, 0.0, 0.999)[: w * h - 7], 0 if expect else 0)
    run(segments, 64, force=64)                                          # (no chunk loop for these: the width is noted and unused)
    # a WRONG width on camera rays: any width is a one-to-one map of the launch's positions onto its rays
    run(image, 64, force=64)
    run(image, 1024, force=1024)
    run(image, 0, force=1004)                                            # (no multiple of 8: unused)
    # (recognised from rays 0, 64, 128 and 256; no whole band of 8 rows: nothing to tile)
    run(image[:300], 256)
    run(image[:200], 0)                                                  # fewer rays than the probes need
    # launches under the shipped threshold take the one-chunk kernel, which maps its chunks the same way (it keeps no statistics)
    gpu.lib().rodent_hip_top_min_rays(-1)
    try:
        for w, h in ((256, 64), (136, 50), (1004, 12), (1920, 33)):
            run(raygen.primary_rays(*cam, w, h, 0.0, 5000.0), None)
        run(image, None, force=64)
        run(image, None, force=0)
        run(segments, None)
        run(image[:200], None)
    finally:
        gpu.lib().rodent_hip_top_min_rays(0)


def test_default_mapping_chooses_chunks_or_refill_by_itself(gpu, oracle, cornell, cornell_dev):
    """BASELINE config 3 ("ray compaction on") without a variant argument: the default kernel (k_bvh2_top_auto) traces rays that share an
    origin as whole chunks and refills idle lanes otherwise (stats[5]: workgroups that chose the refill loop).  With the ray-kind hint ON
    (rodent_hip_ray_kind_hint; off by default) a list -- same pointer, same count -- that was incoherent throughout goes to the refill
    kernel proper from its second launch on (stats[4]), and back when the caller puts coherent rays into the same buffer.  Hits are the
    oracle's in every case."""
    import torch
    top = gpu.variants(2).index("top")
    nodes, tris = cornell.blocks[2]
    n = 9216 * 64 + 77
    base = cornell.ray_sets["primary"]
    coherent = np.tile(base, (n + len(base) - 1) // len(base))[:n].copy()              # one camera
    incoherent = coherent.copy()
    incoherent["org"][:, 1] += (np.arange(n, dtype=np.float32) % 613) * 1e-4
    expected = {id(coherent): oracle.traverse(2, nodes, tris, coherent)[0], id(incoherent): oracle.traverse(2, nodes, tris, incoherent)[0]}
    gpu.lib().rodent_hip_top_min_rays(-1)
    rd = gpu.to_device(incoherent, 0)
    hd = torch.full((n * 16,), 0xFF, dtype=torch.uint8, device="cuda:0")

    def launch(rays):
        hd.fill_(0xFF)
        gpu.read_stats(0)
        gpu.traverse_async(cornell_dev[2], rd, hd, n, False, top)
        gpu.check_errors(0)
        assert gpu.from_device(hd, F.HIT1).tobytes() == expected[id(rays)].tobytes()
        st = gpu.read_stats(0)
        # (the refill kernel ran, the default kernel's waves chose the refill loop)
        return st[4] > 0, st[5] > 0
    try:
        for hint in (True, False):
            gpu.ray_kind_hint(hint)
            rd.copy_(gpu.to_device(incoherent, 0))
            first = launch(incoherent)
            # (a hint left by an earlier list at this address may already apply)
            assert first in ((False, True), (True, False)), first
            assert launch(incoherent) == ((True, False) if hint else (False, True))
            rd.copy_(gpu.to_device(coherent, 0))
            # whichever kernel: the hits are right, and it reports what it saw
            launch(coherent)
            assert launch(coherent) == (False, False)
            assert launch(coherent) == (False, False)
    finally:
        # the shipped default (round 5): kernel selection does not depend on earlier launches
        gpu.ray_kind_hint(False)
        gpu.lib().rodent_hip_top_min_rays(0)
