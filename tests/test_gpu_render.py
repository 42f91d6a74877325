"""GPU parity tests of the wavefront path tracer against the CPU oracle.

Every path's arithmetic is reproduced bit for bit (same RNG stream, same operation order), so ray
counts are EXACT; the film differs only by the order of the floating-point atomic adds (tolerance
1e-5 relative + 1e-6 absolute per channel, stated here)."""
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN
from rodent_amd import scene as S

pytestmark = pytest.mark.gpu
FILM_RTOL, FILM_ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module")
def cornell_scene(native_build, tmp_path_factory):
    return S.convert(GOLDEN / "cornell_box.obj", tmp_path_factory.mktemp("scene") / "cornell.rscene")


@pytest.fixture()
def R(native_build):
    import torch
    from rodent_amd import render
    assert torch.cuda.is_available()
    return render


@pytest.mark.parametrize("spp,max_len,iters", [(1, 64, 2), (4, 64, 3), (4, 0, 1), (3, 2, 2)])
def test_film_matches_oracle(R, oracle, cornell_scene, spp, max_len, iters):
    W, H = 200, 120
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    # with the sort by material (the reference's loop) and without (the default)
    r = R.Renderer(cornell_scene, W, H, spp, max_len, sort=bool(iters % 2))
    film_o = None
    for it in range(iters):
        r.render(cam, it)
        c = r.counters()
        film_o, counts = oracle.render(cornell_scene, cam, it, spp, max_len, W, H, film_o)
        assert (c["primary_rays"], c["shadow_rays"], c["generated"]) == (counts[0], counts[1], W * H * spp)   # exact
    film_g = r.film()
    r.close()
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)
    assert film_g.mean() > 0.02


def test_a_thousand_materials_and_the_per_scene_mapping(R, oracle, cornell_scene):
    """The sort by material with the most bins the reference allows (1024 geometries + the miss bin, mapping_gpu.impala:200,342):
    k_scatter's in-block ranking then asks for 65 600 bytes of dynamic LDS.  And the library's own choice of mapping
    (rodent_hip_render_mapping(dev, -1)): the megakernel for a hierarchy of a few dozen nodes."""
    import copy
    sc = copy.copy(cornell_scene)
    sc.materials = np.ascontiguousarray(np.tile(cornell_scene.materials, 1024 // len(cornell_scene.materials) + 1)[:1024])
    W, H = 150, 90
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    film_o, counts = oracle.render(cornell_scene, cam, 0, 3, 5, W, H)
    for mapping in ("streaming", "auto"):
        r = R.Renderer(sc, W, H, 3, 5, mapping=mapping, sort=True)
        assert r.mapping_name() == ("streaming" if mapping == "streaming" else "megakernel")
        r.render(cam, 0)
        c = r.counters(); film_g = r.film(); r.close()
        assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1])
        assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)


@pytest.mark.parametrize("scene_name", ["cornell", "textured"])
def test_unsorted_shading_traces_the_same_paths(R, oracle, cornell_scene, textured_scene, scene_name):
    """rodent_hip_render_sort(0): no sort by material, the shader ends the rays that missed -- same paths as the oracle."""
    sc = cornell_scene if scene_name == "cornell" else textured_scene[0]
    eye, d = ((0, 1, 2.7), (0, 0, -1)) if scene_name == "cornell" else ((0.3, 1.0, 3.2), (-0.1, -0.25, -1))
    W, H = 180, 110
    cam = S.camera_settings(eye, d, (0, 1, 0), 55, W, H)
    r = R.Renderer(sc, W, H, 3, 7, sort=False, capacity=20000)
    film_o = None
    for it in range(2):
        r.render(cam, it)
        c = r.counters()
        film_o, counts = oracle.render(sc, cam, it, 3, 7, W, H, film_o)
        assert (c["primary_rays"], c["shadow_rays"], c["generated"]) == (counts[0], counts[1], W * H * 3)
    film_g = r.film()
    r.close()
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)


def test_row_bands_reproduce_the_frame(R, oracle, cornell_scene):
    """Tile sharding (multi-GPU path): bands rendered separately sum to the full frame."""
    W, H = 160, 96
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    r = R.Renderer(cornell_scene, W, H, 2, 16)
    r.render(cam, 5)
    full = r.film()
    r.clear()
    for y0, y1 in ((0, 31), (31, 64), (64, 96)):
        r.render_rows(cam, 5, y0, y1)
    bands = r.film()
    r.close()
    assert np.allclose(bands, full, rtol=FILM_RTOL, atol=FILM_ATOL)
    ref, _ = oracle.render(cornell_scene, cam, 5, 2, 16, W, H)
    assert np.allclose(full, ref, rtol=FILM_RTOL, atol=FILM_ATOL)


@pytest.mark.parametrize("capacity", [1 << 20, 100_000, 4096])
def test_capacity_regeneration(R, oracle, cornell_scene, capacity):
    """More paths per frame than a stream holds: the stream is refilled while it drains (mapping_gpu.impala:332-336).
    The reference's capacity (1 Mi), an odd one and a tiny one; the default (32 Mi) is what the other tests run with."""
    W, H, SPP = (640, 480, 5) if capacity == 1 << 20 else (200, 150, 3)       # 1 536 000 / 90 000 paths > capacity
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    r = R.Renderer(cornell_scene, W, H, SPP, 6, capacity=capacity)
    r.render(cam, 0)
    c = r.counters(); film_g = r.film(); r.close()
    film_o, counts = oracle.render(cornell_scene, cam, 0, SPP, 6, W, H)
    assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1]) and c["generated"] == W * H * SPP
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)


def test_lane_refill_on_a_small_tree_matches_oracle(R, oracle, cornell_scene):
    """k_trace_refill where the library would not use it (a 22-node tree: every ray is short, refills follow each other closely):
    640 x 480 x 3 spp in streams of 600 000 rays (the persistent launches need 524 288), joint launch and separate launches, against
    the oracle's film and exact ray counts."""
    W, H, SPP = 640, 480, 3
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    film_o, counts = oracle.render(cornell_scene, cam, 1, SPP, 5, W, H)
    r = R.Renderer(cornell_scene, W, H, SPP, 5, mapping="streaming")
    assert r.trace_refill() == (0, 0); r.close()                             # per scene: off for this one
    for mode, refill in ((2, (32, 32)), (1, (4, 48))):
        r = R.Renderer(cornell_scene, W, H, SPP, 5, mapping="streaming", trace_persistent=mode, trace_refill=refill, capacity=600_000)
        r.render(cam, 1)
        c = r.counters(); film_g = r.film(); r.close()
        assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1]) and c["generated"] == W * H * SPP, (mode, refill)
        assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL), (mode, refill)


def test_cornell_matches_reference_image_full_size(R, cornell_scene):
    """The reference's own CTest (src/CMakeLists.txt:131-134): 1080x720, 50 frames x 4 spp, vs testing/ref-cornell.png."""
    W, H = 1080, 720
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    r = R.Renderer(cornell_scene, W, H, 4, 64)
    for it in range(50):
        r.render(cam, it)
    img = R.tonemap(r.film(), 50).astype(np.float32)
    r.close()
    ref = np.array(Image.open(GOLDEN / "ref-cornell.png").convert("RGB")).astype(np.float32)
    mse = ((img - ref) ** 2).mean() / 255.0 ** 2
    assert mse < 3e-4, mse
    assert np.allclose(img.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=0.02)


def test_rodent_cli(native_build, tmp_path):
    out = tmp_path / "o.png"
    r = subprocess.run([native_build.BIN_DIR / "rodent", "--scene", GOLDEN / "cornell_box.obj", "--bench", "3", "--eye", "0", "1", "2.7",
                        "--dir", "0", "0", "-1", "--up", "0", "1", "0", "--width", "320", "--height", "240", "-o", out],
                       capture_output=True, text=True, check=True)
    assert "(min/med/max Msamples/s)" in r.stdout.strip().splitlines()[-1]        # driver.cpp:344-347
    im = np.array(Image.open(out))
    assert im.shape == (240, 320, 4) and im[..., :3].mean() > 40 and (im[..., 3] == 255).all()
    assert (im[:80, 100:220, :3].min(axis=2) > 250).sum() > 50                     # the ceiling light (top centre) is saturated


def test_rodent_cli_ngpu(native_build, tmp_path):
    """`rodent --ngpu K` (BASELINE config 5's partition in the C++ host): K = 1 is the single-GPU renderer; with as many GPUs
    as the box has, row bands + one RCCL gather reproduce the single-GPU image; more GPUs than the box has is an error, not a crash."""
    import torch
    # (torch first: it brings its own HIP runtime, and that one has to initialise before the library's)
    have = torch.cuda.device_count()
    imgs = {}
    for k in sorted({1, min(have, 2), have}):
        out = tmp_path / f"n{k}.png"
        cmd = [str(native_build.BIN_DIR / "rodent"), "--scene", str(GOLDEN / "cornell_box.obj"), "--eye", "0", "1", "2.7", "--dir", "0",
            "0", "-1", "--up", "0", "1", "0",
               "--width", "200", "--height", "123", "--spp", "4", "--bench", "3", "--target", "amdgpu-streaming", "--ngpu", str(k), "-o",
                   str(out)]
        res = subprocess.run(cmd, capture_output=True, text=True, check=True)
        assert "(min/med/max Msamples/s)" in res.stdout and (k == 1 or f"# GPUs: {k}" in res.stdout)
        imgs[k] = np.asarray(Image.open(out).convert("RGB"), dtype=np.int32)
        assert imgs[k].mean() > 40
    for k, im in imgs.items():
        assert np.abs(im - imgs[1]).max() <= 1, k
    bad = subprocess.run(cmd[:-4] + ["--ngpu", str(have + 1), "-o", str(out)], capture_output=True, text=True)
    assert bad.returncode != 0 and "No such GPU device(s)" in bad.stderr


def test_stage_level_api_reproduces_render(R, oracle, cornell_scene):
    """Drive the wavefront loop from the host through the stage entry points exactly like the reference's
    gpu_streaming_trace (mapping_gpu.impala:308-369) and check stream invariants after every stage:
    sorted by geometry and stable, ray_ends = exclusive ends, dead rays removed in order, film = render()."""
    import ctypes as C
    W, H, SPP, MAXLEN, IT = 96, 64, 2, 5, 1
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    r = R.Renderer(cornell_scene, W, H, SPP, MAXLEN)
    r.render(cam, IT)
    film_ref = r.film(); counters = r.counters()
    r.clear()
    l = R.stage_lib()
    cap = 4096                                                    # small capacity => several refills
    p, q, s = R.PrimaryStream(), R.PrimaryStream(), R.SecondaryStream()
    l.rodent_gpu_get_first_primary_stream(0, C.byref(p), cap)
    l.rodent_gpu_get_second_primary_stream(0, C.byref(q), cap)
    l.rodent_gpu_get_secondary_stream(0, C.byref(s), cap)
    st = R.make_settings(cam)
    G = len(cornell_scene.materials)
    num_rays, ray_id, traced = SPP * W * H, 0, 0
    ends = (C.c_int32 * (G + 1))()
    while ray_id < num_rays or p.size > 0:
        if p.size < cap and ray_id < num_rays:
            n = min(num_rays - ray_id, cap - p.size)
            l.hip_generate_rays(0, C.byref(p), cap, ray_id, n, C.byref(st), IT, W, H, 0, SPP, None)
            ray_id += n
        ids_before = R.read_stream_array(p.rays.id, p.size, "<i4")
        l.hip_traverse_primary(0, C.byref(p), None)
        traced += p.size
        geom_before = R.read_stream_array(p.geom_id, p.size, "<i4")
        rnd_before = R.read_stream_array(p.rnd, p.size, "<u4")
        l.hip_sort_primary(0, C.byref(p), C.byref(q), ends, None)
        p, q = q, p
        geom = R.read_stream_array(p.geom_id, p.size, "<i4")
        assert (np.diff(geom) >= 0).all() and geom.max(initial=0) <= G            # sorted, misses (bin G) last
        assert list(ends) == np.cumsum(np.bincount(geom_before, minlength=G + 1)).tolist()
        order = np.argsort(geom_before, kind="stable")                            # stable: ties keep stream order
        assert np.array_equal(R.read_stream_array(p.rnd, p.size, "<u4"), rnd_before[order])
        assert np.array_equal(R.read_stream_array(p.rays.id, p.size, "<i4"), ids_before[order])
        valid = ends[G - 1]
        l.hip_shade(0, C.byref(p), C.byref(s), valid, None)
        l.hip_traverse_secondary(0, C.byref(s), None)
        alive = R.read_stream_array(p.rays.id, valid, "<i4") >= 0
        depth_before = R.read_stream_array(p.depth, valid, "<i4")
        n_alive = l.hip_compact_primary(0, C.byref(p), C.byref(q), None)
        p, q = q, p
        assert n_alive == alive.sum() == p.size
        assert np.array_equal(R.read_stream_array(p.depth, p.size, "<i4"), depth_before[alive])   # order preserved
        assert (R.read_stream_array(p.rays.id, p.size, "<i4") >= 0).all()
    assert traced == counters["primary_rays"]
    film = r.film()
    r.close()
    assert np.allclose(film, film_ref, rtol=FILM_RTOL, atol=FILM_ATOL)
    ref, _ = oracle.render(cornell_scene, cam, IT, SPP, MAXLEN, W, H)
    assert np.allclose(film, ref, rtol=FILM_RTOL, atol=FILM_ATOL)


@pytest.mark.parametrize("mapping", ["streaming", "megakernel"])
def test_renderer_on_a_hierarchy_deeper_than_the_lds_window(R, oracle, cornell_scene, mapping):
    """The Cornell box under a degenerate CHAIN hierarchy (node k = {triangle k, everything behind it}): rays whose nearer child is
    the rest push a leaf per level, 30 and more entries deep -- past the 15 / 16-entry LDS windows of the stream traversal kernels
    (rays handed to k_trace_deep) and of the megakernel (traced again with its 64-entry LDS + scratch stack).  Same frame as the
    oracle on the same hierarchy."""
    import copy
    from rodent_amd import formats as F
    sc = copy.copy(cornell_scene)
    first = {}
    for i, t in enumerate(cornell_scene.tris):
        first.setdefault(int(t["prim_id"]) & 0x7FFFFFFF, i)
    tris = cornell_scene.tris[sorted(first.values())].copy()
    tris["prim_id"] = (tris["prim_id"].astype(np.int64) | 0x80000000).astype(np.uint32).view(np.int32) if tris[
        "prim_id"].dtype.kind == "i" else tris["prim_id"] | 0x80000000
    n = len(tris)
    # Tri1 stores e1 = v0 - v1, e2 = v2 - v0 (converter.cpp:365-380)
    v0 = tris["v0"].astype(np.float64); v1 = v0 - tris["e1"]; v2 = v0 + tris["e2"]
    # every box is the scene's box (loose bounds are legal): each ray enters both children of every node at the same distance, the
    # strict `<` sends it into the rest first and the leaf goes on the stack -- 35 entries at the bottom of the chain
    lo = np.tile(np.minimum(np.minimum(v0, v1), v2).min(0) - 1e-3, (n, 1)); hi = np.tile(np.maximum(np.maximum(v0, v1), v2).max(0) + 1e-3,
        (n, 1))
    rest_lo, rest_hi = lo, hi
    nodes = np.zeros(n - 1, F.NODE2)
    for k in range(n - 1):
        b = nodes["bounds"][k]
        b[0:6:2], b[1:6:2] = lo[k], hi[k]
        last = k == n - 2
        b[6:12:2], b[7:12:2] = (lo[k + 1], hi[k + 1]) if last else (rest_lo[k + 1], rest_hi[k + 1])
        nodes["child"][k] = (~k, ~(k + 1) if last else k + 2)
    sc.nodes, sc.tris = nodes, tris
    W, H = 120, 80
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    from rodent_amd import raygen
    depth = oracle.ray_depths(nodes, tris, raygen.primary_rays((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H, 0.0, 1e9))
    assert depth.max() >= 30 and (depth > 16).mean() > 0.5                  # really past every LDS window
    film_o, counts = oracle.render(sc, cam, 0, 2, 6, W, H)
    r = R.Renderer(sc, W, H, 2, 6, mapping=mapping)
    r.render(cam, 0)
    c = r.counters(); film_g = r.film(); r.close()
    assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1])
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL) and film_g.mean() > 0.02


def test_stage_level_shader_ends_rays_that_missed(R, cornell_scene):
    """hip_shade on a stream that was NOT sorted first (the reference's loop drops the misses in its sort, mapping_gpu.impala:347-357):
    a ray that missed carries the miss id in geom_id; the shader must end it (id -1, no shadow ray), not index the material
    table with it."""
    import ctypes as C
    W, H = 128, 96
    cam = S.camera_settings((0, 1, 4.5), (0, 0, -1), (0, 1, 0), 80, W, H)          # from outside the box: many rays pass it
    r = R.Renderer(cornell_scene, W, H, 1, 4)
    l = R.stage_lib()
    p, s = R.PrimaryStream(), R.SecondaryStream()
    l.rodent_gpu_get_first_primary_stream(0, C.byref(p), W * H)
    l.rodent_gpu_get_secondary_stream(0, C.byref(s), W * H)
    st = R.make_settings(cam)
    l.hip_generate_rays(0, C.byref(p), W * H, 0, W * H, C.byref(st), 0, W, H, 0, 1, None)
    l.hip_traverse_primary(0, C.byref(p), None)
    geom = R.read_stream_array(p.geom_id, W * H, "<i4")
    missed = geom >= len(cornell_scene.materials)
    assert 0.05 < missed.mean() < 0.95
    l.hip_shade(0, C.byref(p), C.byref(s), W * H, None)
    ids, sids = R.read_stream_array(p.rays.id, W * H, "<i4"), R.read_stream_array(s.rays.id, W * H, "<i4")
    assert (ids[missed] == -1).all() and (sids[missed] == -1).all() and (ids[~missed] >= 0).sum() > 0
    r.close()


@pytest.mark.parametrize("spp,max_len,W,H", [(1, 64, 203, 121), (4, 64, 200, 120), (3, 2, 77, 50), (64, 4, 40, 30), (2048, 1, 5, 3)])
def test_megakernel_matches_oracle(R, oracle, cornell_scene, spp, max_len, W, H):
    """The persistent-threads mapping (mapping_gpu.impala:371-474): same paths, same ray counts, per-path colour sums;
    ragged tiles (film not a multiple of the tile side), spp that is not a power of two, spp > 1024 (tile side 1)."""
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    # both loop structures: k_mega / k_mega_joint
    r = R.Renderer(cornell_scene, W, H, spp, max_len, mapping="megakernel", mega_joint=(W % 2 == 0))
    film_o = None
    for it in range(2):
        r.render(cam, it)
        c = r.counters()
        film_o, counts = oracle.render(cornell_scene, cam, it, spp, max_len, W, H, film_o)
        assert (c["primary_rays"], c["shadow_rays"], c["generated"]) == (counts[0], counts[1], W * H * spp)
    film_g = r.film()
    r.clear()
    for y0, y1 in ((0, H // 3), (H // 3, H)):          # row bands (multi-GPU sharding) through the megakernel
        r.render_rows(cam, 0, y0, y1)
        r.render_rows(cam, 1, y0, y1)
    bands = r.film()
    r.close()
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)
    assert np.allclose(bands, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)


def test_rodent_cli_megakernel_target(native_build, tmp_path):
    """`rodent --target amdgpu-megakernel` renders the same samples as the streaming target: PNGs equal to one 8-bit level."""
    imgs = {}
    for target in ("amdgpu-streaming", "amdgpu-megakernel"):
        out = tmp_path / (target + ".png")
        cmd = [str(native_build.BIN_DIR / "rodent"), "--scene", str(GOLDEN / "cornell_box.obj"), "--eye", "0", "1", "2.7", "--dir", "0",
            "0", "-1",
               "--up", "0", "1", "0", "--width", "300", "--height", "200", "--spp", "8", "--bench", "4", "--target", target, "-o", str(out)]
        res = subprocess.run(cmd, capture_output=True, text=True, check=True)
        assert "(min/med/max Msamples/s)" in res.stdout
        imgs[target] = np.asarray(Image.open(out).convert("RGB"), dtype=np.int32)
    diff = np.abs(imgs["amdgpu-streaming"] - imgs["amdgpu-megakernel"])
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01
    assert imgs["amdgpu-megakernel"].mean() > 40
    bad = subprocess.run(cmd[:-4] + ["--target", "nvvm-streaming"], capture_output=True, text=True)
    assert bad.returncode != 0 and "Unknown target" in bad.stderr


@pytest.mark.parametrize("mapping", ["streaming", "megakernel"])
def test_textured_scene_matches_oracle(R, oracle, textured_scene, mapping):
    """Textures: map_Kd (PNG, JPEG), map_Ks (TGA) with the per-texel diffuse/Phong mix, repeat border,
    bilinear filter -- same paths as the CPU oracle: ray counts exact, film within the usual tolerance."""
    sc, _ = textured_scene
    W, H = 160, 100
    cam = S.camera_settings((0.3, 1.0, 3.2), (-0.1, -0.25, -1), (0, 1, 0), 50, W, H)
    r = R.Renderer(sc, W, H, 4, 8, mapping=mapping)
    film_o = None
    for it in range(2):
        r.render(cam, it)
        c = r.counters()
        film_o, counts = oracle.render(sc, cam, it, 4, 8, W, H, film_o)
        assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1])
    film_g = r.film()
    r.close()
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)
    floor = film_g[70:98, 30:130] / 2
    assert (floor[..., 0] > 1.5 * floor[..., 2]).mean() > 0.1 and (floor[..., 2] > 1.5 * floor[..., 0]).mean() > 0.05


def test_rodent_cli_renders_textured_obj(native_build, textured_scene, tmp_path):
    _, d = textured_scene
    out = tmp_path / "room.png"
    subprocess.run([native_build.BIN_DIR / "rodent", "--scene", d / "room.obj", "--eye", "0.3", "1.0", "3.2", "--dir", "-0.1", "-0.25",
        "-1",
                    "--up", "0", "1", "0", "--fov", "50", "--width", "240", "--height", "160", "--spp", "8", "--bench", "4", "-o", out],
                   capture_output=True, text=True, check=True)
    im = np.asarray(Image.open(out).convert("RGB"), dtype=np.float32)
    floor = im[115:155, 40:200]
    assert (floor[..., 0] > 1.3 * floor[..., 2]).mean() > 0.1 and (floor[..., 2] > 1.3 * floor[..., 0]).mean() > 0.05


def test_stream_traversal_hands_deep_rays_to_the_follow_up_kernel(R, oracle):
    """Stacks deeper than the 16-entry LDS window: k_trace_primary / k_trace_secondary abandon the ray and k_trace_deep
    traces it again with the 64-entry stack in global memory.  A 40-deep chain BVH (stack reaches 40) as the scene of
    the stage-level entry points, deep and shallow rays mixed in one wave: hit fields bit-exact against the oracle,
    and the film gets exactly the unoccluded shadow rays' colours."""
    import ctypes as C
    from types import SimpleNamespace
    from conftest import chain_bvh2
    from rodent_amd import formats as F
    nodes, tris = chain_bvh2(40)
    ntri = len(tris)
    mat = np.zeros(1, S.MATERIAL); mat["kd"] = 0.5; mat["type"] = 1
    light = np.zeros(1, S.LIGHT); light["inv_area"] = 1.0
    scene = SimpleNamespace(vertices=np.zeros((3 * ntri, 4), "<f4"), normals=np.zeros((3 * ntri, 4), "<f4"),
        face_normals=np.zeros((ntri, 4), "<f4"),
                            indices=np.zeros((ntri, 4), "<i4"), nodes=nodes, tris=tris, materials=mat, lights=light,
                                light_ids=np.zeros(ntri, "<i4"),
                            texcoords=np.zeros((0, 4), "<f4"), textures=np.zeros(0, S.TEXTURE), texels=np.zeros(0, "<u4"), num_tris=ntri)
    n = 1000
    rng = np.random.default_rng(3)
    org = np.zeros((n, 3), "<f4"); org[:, :2] = rng.uniform(-4, 4, (n, 2)); org[::3, 0] += 50.0          # every 3rd ray misses everything
    d = np.tile(np.float32([0.001, 0.002, 1.0]), (n, 1))
    rays = F.make_rays(org, d, 0.0, 1000.0)
    W, H = 40, 25                                                 # film of n pixels: ray k is pixel k
    r = R.Renderer(scene, W, H, 1, 4)
    l = R.stage_lib()
    p, s = R.PrimaryStream(), R.SecondaryStream()
    l.rodent_gpu_get_first_primary_stream(0, C.byref(p), n)
    l.rodent_gpu_get_secondary_stream(0, C.byref(s), n)
    for stream_rays in (p.rays, s.rays):
        R.write_stream_array(stream_rays.id, np.arange(n, dtype="<i4"))
        for k, name in enumerate(("org_x", "org_y", "org_z")):
            R.write_stream_array(getattr(stream_rays, name), rays["org"][:, k])
        for k, name in enumerate(("dir_x", "dir_y", "dir_z")):
            R.write_stream_array(getattr(stream_rays, name), rays["dir"][:, k])
        R.write_stream_array(stream_rays.tmin, rays["tmin"]); R.write_stream_array(stream_rays.tmax, rays["tmax"])
    color = rng.uniform(0.1, 1.0, (n, 3)).astype("<f4")
    for k, name in enumerate(("color_r", "color_g", "color_b")):
        R.write_stream_array(getattr(s, name), color[:, k])
    p.size = n; s.size = n
    ref, st = oracle.traverse(2, nodes, tris, rays)
    assert st["max_stack"] == 40 and (ref["tri_id"] >= 0).sum() > 600
    l.hip_traverse_primary(0, C.byref(p), None)
    got_prim = R.read_stream_array(p.prim_id, n, "<i4")
    assert np.array_equal(got_prim, ref["tri_id"])
    for name in ("t", "u", "v"):
        got = R.read_stream_array(getattr(p, name), n, "<f4")
        hit = ref["tri_id"] >= 0
        assert got[hit].tobytes() == ref[name][hit].tobytes() and (name != "t" or got[~hit].tobytes() == ref["t"][~hit].tobytes())
    assert np.array_equal(R.read_stream_array(p.geom_id, n, "<i4"), np.where(ref["tri_id"] >= 0, 0, 1))
    r.clear()
    l.hip_traverse_secondary(0, C.byref(s), None)
    film = r.film().reshape(-1, 3)
    occluded, _ = oracle.traverse(2, nodes, tris, rays, any_hit=True)
    expect = np.where((occluded["tri_id"] < 0)[:, None], color, 0.0).astype("<f4")
    assert np.array_equal(film, expect)
    r.close()


@pytest.mark.parametrize("mapping,sort,overlap", [("streaming", True, True), ("streaming", False, True), ("streaming", True, False),
                                                  ("streaming", False, False), ("megakernel", True, True)])
def test_every_bsdf_matches_oracle(R, oracle, materials_scene, mapping, sort, overlap):
    """Diffuse, Phong, mix, mirror, glass (refraction and total internal reflection inside a slab), black and an emitter in
    one room: the GPU shader takes the same paths as the oracle -- ray counts exact, film within tolerance."""
    W, H = 150, 100
    cam = S.camera_settings((0, 1, 2.6), (0, -0.05, -1), (0, 1, 0), 60, W, H)
    # 60 000 paths: refills too
    r = R.Renderer(materials_scene, W, H, 4, 12, mapping=mapping, sort=sort, overlap=overlap, capacity=30000)
    film_o = None
    for it in range(2):
        r.render(cam, it)
        c = r.counters()
        film_o, counts = oracle.render(materials_scene, cam, it, 4, 12, W, H, film_o)
        assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1])
    film_g = r.film()
    r.close()
    assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL)


def test_device_film_view_aliases_the_library_film(R, cornell_scene):
    """parallel.device_film: a zero-copy torch view of the DEVICE film (rodent_get_film_data) -- what the multi-GPU path hands
    to RCCL (gather_film_to_root) instead of bouncing the band through the host."""
    import torch
    from rodent_amd import parallel
    W, H = 64, 40
    cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
    r = R.Renderer(cornell_scene, W, H, 2, 4)
    r.render(cam, 0)
    view = parallel.device_film(0)
    assert view.is_cuda and tuple(view.shape) == (H, W, 3)
    assert np.array_equal(view.cpu().numpy(), r.film())
    r.render(cam, 1)                                             # the view follows the film: it is the same memory
    assert np.array_equal(view.cpu().numpy(), r.film())
    assert parallel.gather_film_to_root(view, None) is view                                   # single process: no collective
    r.close()


def test_mid_size_textured_scene_gets_the_rules_it_should(R, oracle, textured_hall):
    """The per-scene rules on a third kind of scene: a textured hall of ~9 500 triangles, a few thousand BVH nodes -- far above
    the 128 nodes up to which the megakernel is chosen (render.hip resolve_mapping), below the 16 384 from which the traversal launches
    refill idle lanes (resolve_refill).  The library must choose the streaming loop without lane refill, both mappings must trace the
    oracle's paths through the textures, and the choice must not lose to the alternative by more than the run-to-run spread."""
    import time, torch
    sc = textured_hall
    assert 128 < len(sc.nodes) < 16384, len(sc.nodes)
    W, H = 192, 120
    cam = S.camera_settings((0.3, 1.0, 3.2), (-0.1, -0.25, -1), (0, 1, 0), 55, W, H)
    film_o, counts = oracle.render(sc, cam, 1, 3, 7, W, H)
    for mapping in ("auto", "streaming", "megakernel"):
        r = R.Renderer(sc, W, H, 3, 7, mapping=mapping)
        if mapping == "auto":
            assert r.mapping_name() == "streaming" and r.trace_refill() == (0, 0)
        r.render(cam, 1)
        c = r.counters(); film_g = r.film(); r.close()
        assert (c["primary_rays"], c["shadow_rays"]) == (counts[0], counts[1]), mapping
        assert np.allclose(film_g, film_o, rtol=FILM_RTOL, atol=FILM_ATOL), mapping
    ms = {}
    # the rule against the measurement, at a frame size where the mappings differ
    for mapping in ("streaming", "megakernel"):
        r = R.Renderer(sc, 1280, 720, 16, 8, mapping=mapping)
        cam2 = S.camera_settings((0.3, 1.0, 3.2), (-0.1, -0.25, -1), (0, 1, 0), 55, 1280, 720)
        r.render_rows(cam2, 0, 0, 720)
        t = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.render_rows(cam2, it, 0,
                720); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        ms[mapping] = min(t) * 1e3
        r.close()
    assert ms["streaming"] <= 1.15 * ms["megakernel"], ms


def test_shading_through_indices_and_normals_still_matches_oracle(native_build):
    """The shader reads a hit's face normal and vertex normals from ONE gathered 48-byte record per triangle (SceneDev::tri_shade, built at
    scene creation); RODENT_HIP_TRI_SHADE=0 keeps the reference's path through indices -> normals (geometry.impala:21-54).  The switch is
    read once per process, so the oracle comparisons of this module run again in a process that has it off."""
    import os, sys
    from conftest import ROOT
    env = dict(os.environ, RODENT_HIP_TRI_SHADE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_render.py"), "-m", "gpu", "-q", "-x", "-k",
                        "test_film_matches_oracle or test_textured_scene_matches_oracle or test_every_bsdf_matches_oracle or "
                            "test_megakernel_matches_oracle"],
                       capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_miss_records_up_front_still_match_oracle(native_build):
    """Round 6: k_trace_refill stores a closest-hit ray's miss record when the ray retires without an accepted triangle (the default);
    RODENT_HIP_LAZY_MISS=0 keeps round 5's record up front.  The switch is read once per process, so the film / ray-count comparisons of
    this module and the scene-class frames run again in a process that has it off."""
    import os, sys
    from conftest import ROOT
    env = dict(os.environ, RODENT_HIP_LAZY_MISS="0")
    files = [str(ROOT / "tests" / "test_gpu_render.py"), str(ROOT / "tests" / "test_gpu_atrium.py")]
    which = "test_film_matches_oracle or test_textured_scene_matches_oracle or test_capacity_regeneration or " \
        "test_atrium_compaction_modes_match_oracle"
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-k", which], capture_output=True, text=True,
        cwd=ROOT, env=env)
    import re
    passed = re.search(r"(\d+) passed", r.stdout)
    # (16 cases: Cornell + atrium)
    assert r.returncode == 0 and passed and int(passed.group(1)) >= 12, r.stdout[-2000:] + r.stderr[-2000:]
