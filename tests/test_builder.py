"""CPU tests for the host-side BVH builder / layout writers (bvh_extractor)."""
import subprocess

import numpy as np
import pytest

from rodent_amd import formats as F


def tri_vertices(v0, e1, e2):
    return v0, v0 - e1, v0 + e2           # e1 = v0 - v1, e2 = v2 - v0


def check_bvh2(nodes, tris, num_prims):
    seen = np.zeros(num_prims, bool)
    visited_tris = np.zeros(len(tris), bool)
    stack = [(1, None)]
    while stack:
        nid, box = stack.pop()
        nd = nodes[nid - 1]
        for k in range(2):
            c = int(nd["child"][k])
            b = nd["bounds"][6 * k: 6 * k + 6]
            if c == 0:
                assert np.isinf(b).all()                      # converter.cpp:343-350
                continue
            lo, hi = b[[0, 2, 4]], b[[1, 3, 5]]
            assert (lo <= hi).all()
            if box is not None:                               # children are inside the parent slot
                assert (lo >= box[0] - 1e-4).all() and (hi <= box[1] + 1e-4).all()
            if c > 0:
                stack.append((c, (lo, hi)))
            else:
                j = ~c
                while True:
                    t = tris[j]
                    visited_tris[j] = True
                    pid = int(t["prim_id"]) & 0x7FFFFFFF
                    seen[pid] = True
                    # the (possibly clipped) reference must overlap the leaf box
                    vs = np.stack(tri_vertices(t["v0"], t["e1"], t["e2"]))
                    assert (vs.max(0) >= lo - 1e-3).all() and (vs.min(0) <= hi + 1e-3).all()
                    j += 1
                    if int(t["prim_id"]) < 0:
                        break
    assert seen.all(), "every primitive must be referenced by some leaf"
    assert visited_tris.all(), "no orphan Tri1 records"


def check_wide(nodes, tris, arity, num_prims):
    seen = np.zeros(num_prims, bool)
    stack = [1]
    while stack:
        nd = nodes[stack.pop() - 1]
        children = nd["child"]
        nz = np.nonzero(children == 0)[0]
        first_empty = nz[0] if len(nz) else arity
        assert (children[first_empty:] == 0).all()            # packed from slot 0 (mapping_cpu.impala:333-334)
        assert np.isposinf(nd["bounds"][0::2, first_empty:]).all() and np.isneginf(nd["bounds"][1::2, first_empty:]).all()
        for k in range(first_empty):
            c = int(children[k])
            if c > 0:
                stack.append(c)
            else:
                j = ~c
                while True:
                    p = tris[j]
                    ids = p["prim_id"]
                    valid = ids != -1
                    assert valid[0] and (np.diff(valid.astype(int)) <= 0).all()   # valid lanes first
                    seen[ids[valid] & 0x7FFFFFFF] = True
                    e1, e2, n = p["e1"][:, valid], p["e2"][:, valid], p["n"][:, valid]
                    assert np.allclose(np.cross(e1.T, e2.T), n.T, rtol=1e-5, atol=1e-7)   # n = e1 x e2 (converter.cpp:229)
                    j += 1
                    if int(ids[3]) < 0:
                        break
    assert seen.all()


def test_cornell_blocks_are_valid(cornell):
    check_bvh2(*cornell.blocks[2], 36)
    check_wide(*cornell.blocks[4], 4, 36)
    check_wide(*cornell.blocks[8], 8, 36)
    # geom_id carries the material index (converter.cpp:246,374): Cornell has 8 materials + default
    g = cornell.blocks[2][1]["geom_id"]
    assert g.min() >= 1 and g.max() <= 8


def write_obj(path, verts, faces):
    with open(path, "w") as f:
        for v in verts:
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for a, b, c in faces:
            f.write(f"f {a + 1} {b + 1} {c + 1}\n")


def test_single_triangle_scene(tmp_path, native_build, oracle):
    """A single-leaf scene still gets a root node whose second slot is empty (bvh.h:218-224)."""
    write_obj(tmp_path / "one.obj", [(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0, 1, 2)])
    subprocess.run([native_build.BIN_DIR / "bvh_extractor", "-obj", tmp_path / "one.obj", "-o", tmp_path / "one.bvh"], check=True,
        stdout=subprocess.DEVNULL)
    n2, t1 = F.read_bvh(tmp_path / "one.bvh", F.BVH2_TRI1)
    assert len(n2) == 1 and n2["child"][0][1] == 0 and n2["child"][0][0] == ~0
    # (no exactly-zero direction components: the reference's octant/safe_rcp combination mishandles those)
    rays = F.make_rays([[0.2, 0.2, 1.0], [2.0, 2.0, 1.0]], [[0.002, 0.001, -1.0], [0.002, 0.001, -1.0]], 0.0, 10.0)
    for w, b in ((2, F.BVH2_TRI1), (4, F.BVH4_TRI4), (8, F.BVH8_TRI4)):
        n, t = F.read_bvh(tmp_path / "one.bvh", b)
        h, _ = oracle.traverse(w, n, t, rays)
        assert h["tri_id"].tolist() == [0, -1] and abs(h["t"][0] - 1.0) < 1e-5


def test_spatial_splits_on_mixed_sizes(tmp_path, native_build, oracle):
    """A carpet of small triangles under a few scene-sized slanted ones: the builder must
    duplicate references of the big triangles (spatial splits, Stich et al.) and the result
    must still be a correct hierarchy."""
    rng = np.random.default_rng(3)
    verts, faces = [], []
    for i in range(24):
        for j in range(24):
            c = np.array([i, 0.0, j]) + rng.uniform(-0.2, 0.2, 3)
            verts += [c, c + [0.4, 0.05, 0.0], c + [0.0, 0.05, 0.4]]
            faces.append((len(verts) - 3, len(verts) - 2, len(verts) - 1))
    for k in range(4):                                     # huge slanted triangles crossing everything
        verts += [np.array([-1.0, 0.1 + k, -1.0]), np.array([25.0, 0.3 + k, -1.0 + k]), np.array([-1.0 + k, 0.5 + k, 25.0])]
        faces.append((len(verts) - 3, len(verts) - 2, len(verts) - 1))
    num = len(faces)
    write_obj(tmp_path / "mixed.obj", verts, faces)
    out = subprocess.run([native_build.BIN_DIR / "bvh_extractor", "-obj", tmp_path / "mixed.obj", "-o", tmp_path / "mixed.bvh"],
                         check=True, capture_output=True, text=True).stdout
    n2, t1 = F.read_bvh(tmp_path / "mixed.bvh", F.BVH2_TRI1)
    assert len(t1) > num, out                                   # duplicated references
    check_bvh2(n2, t1, num)
    n8, t4 = F.read_bvh(tmp_path / "mixed.bvh", F.BVH8_TRI4)
    check_wide(n8, t4, 8, num)
    org = rng.uniform(-1, 25, (2000, 3)); org[:, 1] = rng.uniform(3, 6, 2000)
    dst = rng.uniform(-1, 25, (2000, 3)); dst[:, 1] = rng.uniform(-2, 0, 2000)
    rays = F.make_rays(org, dst - org, 0.0, 1.0)
    brute, _ = oracle.brute_force(t1, rays)
    for w, (n, t) in ((2, (n2, t1)), (8, (n8, t4))):
        h, st = oracle.traverse(w, n, t, rays)
        assert np.array_equal(h["tri_id"] >= 0, brute["tri_id"] >= 0)
        hit = brute["tri_id"] >= 0
        assert hit.sum() > 200 and np.allclose(h["t"][hit], brute["t"][hit], rtol=1e-4)
        assert st["max_stack"] < 64


def test_no_spatial_flag_gives_one_ref_per_triangle(tmp_path, native_build, cornell):
    subprocess.run([native_build.BIN_DIR / "bvh_extractor", "-obj", cornell.bvh_path.parent / "cornell_box.obj", "-o", tmp_path / "c.bvh",
        "--no-spatial"],
                   check=True, stdout=subprocess.DEVNULL)
    _, t1 = F.read_bvh(tmp_path / "c.bvh", F.BVH2_TRI1)
    assert len(t1) == 36


def test_obj_without_faces_is_an_error_not_a_crash(tmp_path, native_build):
    """A mesh with vertices but no faces used to make one empty leaf and out.back() on an empty vector (exit 139)."""
    (tmp_path / "empty.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\n")
    for tool, args in (("bvh_extractor", ["-obj", tmp_path / "empty.obj", "-o", tmp_path / "e.bvh"]),
        ("converter", [tmp_path / "empty.obj", "-o", tmp_path / "e.rscene"])):
        r = subprocess.run([native_build.BIN_DIR / tool, *args], capture_output=True, text=True)
        assert r.returncode == 1 and "no faces" in r.stderr, (tool, r.returncode, r.stderr)


def test_converter_builder_parameters(tmp_path, native_build, oracle):
    """converter --bvh-leaf / --bvh-traversal-cost (the sweep of DESIGN 3.4.2): larger leaves and a dearer inner node give smaller
    hierarchies that are still valid BVH2 blocks and still trace to the exhaustive checker's hits; the defaults are the reference's
    parameters (2 references, cost 1)."""
    from conftest import ROOT, write_textured_hall
    from rodent_amd import build, raygen, scene as S
    obj = write_textured_hall(tmp_path)
    scenes = {}
    for name, args in (("default", []), ("explicit", ["--bvh-leaf", "2", "--bvh-traversal-cost", "1"]), ("leaf8", ["--bvh-leaf", "8"]),
        ("cost3", ["--bvh-traversal-cost", "3"])):
        out = tmp_path / f"{name}.rscene"
        subprocess.run([str(build.BIN_DIR / "converter"), str(obj), "-o", str(out), *args], check=True, stdout=subprocess.DEVNULL)
        scenes[name] = S.Scene(out)
    assert scenes["default"].nodes.tobytes() == scenes["explicit"].nodes.tobytes() and scenes["default"].tris.tobytes() == scenes[
        "explicit"].tris.tobytes()
    assert len(scenes["leaf8"].nodes) < 0.5 * len(scenes["default"].nodes) and len(scenes["cost3"].nodes) < len(scenes["default"].nodes)
    b = np.asarray(scenes["default"].nodes["bounds"][0]).reshape(2, 6)
    lo, hi = np.minimum(b[0, 0::2], b[1, 0::2]), np.maximum(b[0, 1::2], b[1, 1::2])
    rays = raygen.random_rays(lo, hi, 4096, 7, 0.0, 1.0)
    ref, _ = oracle.traverse(2, scenes["default"].nodes, scenes["default"].tris, rays)
    for name in ("leaf8", "cost3"):
        sc = scenes[name]
        check_bvh2(sc.nodes, sc.tris, sc.num_tris)
        got, _ = oracle.traverse(2, sc.nodes, sc.tris, rays)
        assert np.array_equal(got["tri_id"] >= 0, ref["tri_id"] >= 0)
        hit = ref["tri_id"] >= 0
        # another hierarchy: the same surfaces (ties may name another triangle)
        assert np.allclose(got["t"][hit], ref["t"][hit], rtol=1e-5, atol=0)


def test_stress_scenes_get_emissive_panels_for_the_renderer(native_build, tmp_path):
    """scene_gen's crown and plant are geometry only; scenes.scene_obj appends emissive panels (material "light", facing down) so that the
    renderer has something that emits -- the converter must find them as lights, and the panels must not touch the .bvh route of the
    traversal matrix."""
    from rodent_amd import scene as S, scenes
    for kind, panels in scenes.PANELS.items():
        obj = scenes.scene_obj(f"{kind}/1")
        text = obj.read_text().splitlines()
        tail = [l for l in text[-(6 * len(panels) + 1):]]
        assert tail[0] == "usemtl light" and sum(l.startswith("f ") for l in tail) == 2 * len(panels)
        sc = S.convert(obj, tmp_path / f"{kind}.rscene")
        assert len(sc.lights) == 2 * len(panels)
