"""The other scene classes of the reference's benchmark suite (benchmarks/benchmark.py:16-44: crown, san-miguel, powerplant beside sponza)
and its third ray class, at test size: seeded stand-ins (rodent_amd/host/stress_scenes.cpp, atrium.cpp at a higher detail) built by the
in-tree builder where the test runs -- a dense organic surface (crown/1: 262 K small triangles), a hall of long thin triangles (plant/1: 520
K triangles, 3.6 M references after spatial splits, tree depth 30+) and the atrium at four times its triangle count (gallery/2) -- traced
with 256 Ki camera rays, 256 Ki random segments and 256 Ki "ao" rays (ray_gen shadow, tools/ray_gen/ray_gen.cpp:60-85: from a point light to
the camera rays' hit points, any hit, tmax 0.999).  Every shipped BVH2 mapping: the whole Hit1 record of every ray bit for bit against
oracle B1 (any hit: the oracle's record as well -- same visit order).  Full-size figures: scripts/scene_matrix.py,
profiles/r05_scene_matrix.txt, bench.py extra.scenes."""
import numpy as np
import pytest

from rodent_amd import formats as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(native_build):
    import torch
    from rodent_amd import abi
    assert torch.cuda.is_available(), "these tests need a GPU"
    abi.lib()
    return abi


@pytest.mark.parametrize("scene", ["crown/1", "plant/1", "gallery/2"])
def test_scene_classes_bit_exact(gpu, oracle, scene):
    from rodent_amd import raygen, scenes
    path = scenes.scene_bvh(scene)
    nodes, tris = F.read_bvh(path, F.BVH2_TRI1)
    assert len(nodes) > 100000
    bvh = gpu.DeviceBvh(2, nodes, tris, 0)
    kind = scene.split("/")[0]
    lo, hi = raygen.scene_bounds2(nodes)
    prim = raygen.primary_rays(*scenes.CAMERAS[kind], 512, 512, 0.0, scenes.PRIMARY_TMAX)
    ref_p, st = oracle.traverse(2, nodes, tris, prim)
    assert st["max_stack"] < 64 and (ref_p["tri_id"] >= 0).mean() > 0.3
    sets = {"primary": (prim, False, ref_p), "random": (raygen.random_rays(lo, hi, 1 << 18, 42, 0.0, scenes.RANDOM_TMAX), False, None),
            "ao": (raygen.shadow_rays(scenes.LIGHTS[kind], prim, ref_p["t"], 0.0, 0.999), True, None)}
    # 256 Ki rays are below the switch point: send the default mapping through its persistent kernel as well
    gpu.lib().rodent_hip_top_min_rays(0)
    try:
        for name, (rays, any_hit, ref) in sets.items():
            if ref is None:
                ref, _ = oracle.traverse(2, nodes, tris, rays, any_hit=any_hit)
            assert 0.05 < (ref["tri_id"] >= 0).mean() <= 1.0, (scene, name)
            for v in gpu.order_preserving_variants(2):
                got = gpu.traverse(bvh, rays, any_hit=any_hit, variant=v)
                bad = np.nonzero((got.view("<u4").reshape(-1, 4) != ref.view("<u4").reshape(-1, 4)).any(axis=1))[0]
                assert len(bad) == 0, f"{scene} {name} {gpu.variants(2)[v]}: {len(bad)} rays differ, first {bad[0]}: {got[bad[0]]} vs " \
                    f"{ref[bad[0]]}"
    finally:
        gpu.lib().rodent_hip_top_min_rays(-1)
    gpu.check_errors(0)


@pytest.mark.parametrize("scene", ["crown/1", "plant/1"])
@pytest.mark.parametrize("mapping", ["auto", "megakernel"])
def test_stress_scenes_path_traced_match_oracle(native_build, oracle, scene, mapping, tmp_path):
    """The renderer on the two stress scenes (lit by the panels scenes.PANELS appends to their OBJ: the generators make geometry only): a
    small frame through the library's own choice of mapping and through the megakernel against the render oracle -- ray counts exact, film
    within the order of the atomic adds.  The full-size frames are measured by scripts/refill_rule_check.py
    (profiles/r05_refill_rule_check.txt)."""
    import torch
    from rodent_amd import render as R, scene as S, scenes
    assert torch.cuda.is_available()
    sc = S.convert(scenes.scene_obj(scene), tmp_path / "scene.rscene")
    assert len(sc.lights) >= 2 and len(sc.nodes) > 100000
    W, H, SPP, MAXLEN, IT = 160, 90, 2, 6, 2
    cam = S.camera_settings(*scenes.CAMERAS[scene.split("/")[0]], W, H)
    film_o, counts = oracle.render(sc, cam, IT, SPP, MAXLEN, W, H, threads=32)
    assert film_o.mean() > 1e-4 and counts[1] > 0                                      # lit: shadow rays were cast and some arrived
    r = R.Renderer(sc, W, H, SPP, MAXLEN, mapping=mapping)
    r.render(cam, IT)
    c = r.counters(); film_g = r.film(); r.close()
    assert (c["primary_rays"], c["shadow_rays"], c["generated"]) == (counts[0], counts[1], W * H * SPP)
    assert np.allclose(film_g, film_o, rtol=1e-5, atol=1e-6)
