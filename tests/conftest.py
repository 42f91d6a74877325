import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def native_build():
    """Host tools + oracle built once per session (hipcc cross-compiles without a GPU)."""
    from rodent_amd import build
    build.build_host()
    build.build_oracle()
    build.build_hip_lib()
    build.build_hip_tools()
    return build


@pytest.fixture(scope="session")
def oracle(native_build):
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def cornell():
    """Committed Cornell-box fixture: BVH blocks, ray sets and oracle outputs."""
    from rodent_amd import formats as F
    sys.path.insert(0, str(GOLDEN))
    import make_golden as mg

    class Fixture:
        bvh_path = GOLDEN / "cornell.bvh"
        blocks = {w: F.read_bvh(GOLDEN / "cornell.bvh", b) for w, b in ((2, F.BVH2_TRI1), (4, F.BVH4_TRI4), (8, F.BVH8_TRI4))}
        ray_sets = {name: F.read_rays(GOLDEN / rf, tmin, tmax) for name, (rf, tmin, tmax) in mg.RAY_SETS.items()}
        expected = dict(np.load(GOLDEN / "cornell-expected.npz"))
        algos = mg.ALGOS
    return Fixture


def ambiguous_mask(brute_hits, second_t, rel=1e-4):
    """Rays whose winning primitive is order-dependent: another primitive has a t within
    `rel` of the closest one (shared edges, duplicated faces).  The reference's own
    variants disagree on those (tools/CMakeLists.txt:24-25, intersection.impala:181-182)."""
    t = brute_hits["t"]
    return (brute_hits["tri_id"] >= 0) & (np.abs(second_t - t) <= rel * np.maximum(np.abs(t), 1e-30))


def chain_bvh2(depth, z0=200.0):
    """Hand-made degenerate BVH2 whose traversal stack grows to `depth` entries for a ray along +z:
    node i = {child0: inner chain (box z in [1,100], entered first), child1: leaf i (triangle at z0+i)}.
    Every level pushes its leaf, so the stack holds `depth` entries before the first pop; the leaves
    then pop far-to-near and each one is accepted, ending at triangle 0 (t = z0)."""
    from rodent_amd import formats as F
    nodes = np.zeros(depth, F.NODE2)
    tris = np.zeros(depth + 1, F.TRI1)
    for i in range(depth + 1):
        z = z0 + i
        v0, v1, v2 = np.float32([-10, -10, z]), np.float32([30, -10, z]), np.float32([-10, 30, z])
        tris[i]["v0"] = v0; tris[i]["e1"] = v0 - v1; tris[i]["e2"] = v2 - v0
        tris[i]["prim_id"] = np.int32(i) | np.int32(-2 ** 31)          # one triangle per leaf
    for i in range(depth):
        last = i == depth - 1
        inner = [-5, 5, -5, 5, 1, 100] if not last else [-5, 5, -5, 5, z0 + depth, z0 + depth]
        leaf = [-5, 5, -5, 5, z0 + i, z0 + i]
        nodes[i]["bounds"] = inner + leaf
        nodes[i]["child"] = [(i + 2) if not last else ~depth, ~i]
    return nodes, tris


def pad_bvh2_depth(nodes, levels):
    """The same hierarchy under `levels` extra nodes above its root, each of which leaves ONE entry on the stack of every ray
    that enters the scene box: wrapper i = {child0: a dummy inner node with two empty slots, child1: the next wrapper (the last
    one: the old root)}, both with the scene box, so the two entry distances are equal, the strict `<` of mapping_gpu.impala:128
    sends the ray into child1 and child0 is pushed.  A ray's deepest stack grows by `levels` entries, its hit does not change;
    it pays 2 x levels extra node steps (the wrappers, and the dummies popped at the very end)."""
    from rodent_amd import formats as F
    b = nodes[0]["bounds"]
    box = [min(b[0], b[6]), max(b[1], b[7]), min(b[2], b[8]), max(b[3], b[9]), min(b[4], b[10]), max(b[5], b[11])]
    shift = levels + 1
    out = np.zeros(len(nodes) + shift, F.NODE2)
    out[shift:] = nodes
    child = out["child"][shift:]
    child[child > 0] += shift                                           # inner ids are 1-based indices; leaves (~first triangle) stay
    inf = np.float32(np.inf)
    for i in range(levels):
        out[i]["bounds"] = box + box
        # dummy (index `levels`), next wrapper / old root (index levels + 1)
        out[i]["child"] = [levels + 1, i + 2 if i + 1 < levels else levels + 2]
    out[levels]["bounds"] = [inf, -inf] * 6
    out[levels]["child"] = [0, 0]
    return out


def write_textured_scene(d):
    """A small textured room for the texture tests: floor with a PNG checker map_Kd, back wall with a JPEG map_Kd AND a
    TGA map_Ks (diffuse/Phong mix whose weight varies per texel), a plain red side wall and a ceiling light.  Texture
    coordinates exceed [0, 1] (repeat border) and are flipped on one quad."""
    import numpy as np
    from PIL import Image
    d.mkdir(parents=True, exist_ok=True)
    yy, xx = np.mgrid[0:32, 0:32]
    checker = np.where(((xx // 4 + yy // 4) % 2)[..., None] == 0, np.array([230, 230, 40], np.uint8),
        np.array([30, 60, 200], np.uint8)).astype(np.uint8)
    checker[:4, :4] = (255, 0, 0)                                     # orientation marker: top-left of the file
    Image.fromarray(checker, "RGB").save(d / "checker.png")
    grad = np.stack([xx * 8, yy * 8, 255 - xx * 4], -1).astype(np.uint8)
    Image.fromarray(grad, "RGB").resize((48, 24), Image.BILINEAR).save(d / "grad.jpg", quality=95, subsampling=0)
    spec = ((xx % 8 < 4) * 200).astype(np.uint8)
    Image.fromarray(np.stack([spec, spec, spec], -1), "RGB").save(d / "spec.tga")
    (d / "room.mtl").write_text(
        "newmtl floor\nKd 1 1 1\nmap_Kd checker.png\n"
        "newmtl back\nKd 0.5 0.5 0.5\nKs 0.3 0.3 0.3\nNs 40\nmap_Kd grad.jpg\nmap_Ks spec.tga\n"
        "newmtl side\nKd 0.8 0.1 0.1\n"
        "newmtl lamp\nKd 0 0 0\nKe 12 12 12\n")
    (d / "room.obj").write_text(
        "mtllib room.mtl\n"
        "v -1 0 1\nv 1 0 1\nv 1 0 -1\nv -1 0 -1\n"            # floor 1-4
        "v -1 0 -1\nv 1 0 -1\nv 1 2 -1\nv -1 2 -1\n"          # back 5-8
        "v -1 0 1\nv -1 0 -1\nv -1 2 -1\nv -1 2 1\n"          # side 9-12
        "v -0.4 1.98 0.4\nv 0.4 1.98 0.4\nv 0.4 1.98 -0.4\nv -0.4 1.98 -0.4\n"   # lamp 13-16
        "vt 0 0\nvt 2.5 0\nvt 2.5 2.5\nvt 0 2.5\n"            # floor: repeats 2.5 times
        "vt 1 0\nvt 0 0\nvt 0 1\nvt 1 1\n"                    # back: mirrored in u
        "vt -0.25 -0.25\n"
        "usemtl floor\nf 1/1 2/2 3/3\nf 1/1 3/3 4/4\n"
        "usemtl back\nf 5/5 6/6 7/7\nf 5/5 7/7 8/8\n"
        "usemtl side\nf 9/9 10/9 11/9\nf 9/9 11/9 12/9\n"
        "usemtl lamp\nf 13/9 15/9 14/9\nf 13/9 16/9 15/9\n")
    return d / "room.obj"


@pytest.fixture(scope="session")
def textured_scene(native_build, tmp_path_factory):
    from rodent_amd import scene as S
    d = tmp_path_factory.mktemp("textured")
    obj = write_textured_scene(d)
    return S.convert(obj, d / "room.rscene"), d


def write_textured_hall(d, floor_cells=56, wall_cells=40):
    """A MID-SIZE textured scene for the per-scene rules (a few thousand BVH nodes: between the Cornell box's dozen and the atrium's
    142 444): the textured room's materials on a bumpy floor of floor_cells^2 quads (PNG checker map_Kd), a rippled back wall of
    wall_cells^2 quads (JPEG map_Kd + TGA map_Ks), a plain side wall and a ceiling light."""
    import numpy as np
    write_textured_scene(d)                                           # textures + room.mtl
    v, vt, f = [], [], {"floor": [], "back": [], "side": [], "lamp": []}

    def grid(name, n, point, uv):
        base = len(v)
        for j in range(n + 1):
            for i in range(n + 1):
                a, b = i / n, j / n
                v.append(point(a, b)); vt.append(uv(a, b))
        for j in range(n):
            for i in range(n):
                p = base + j * (n + 1) + i + 1
                f[name] += [(p, p + 1, p + n + 2), (p, p + n + 2, p + n + 1)]
    grid("floor", floor_cells, lambda a, b: (-1 + 2 * a, 0.04 * np.sin(9 * a) * np.cos(7 * b), 1 - 2 * b), lambda a, b: (2.5 * a, 2.5 * b))
    grid("back", wall_cells, lambda a, b: (-1 + 2 * a, 2 * b, -1 + 0.03 * np.sin(11 * a + 5 * b)), lambda a, b: (1 - a, b))
    grid("side", 1, lambda a, b: (-1, 2 * b, 1 - 2 * a), lambda a, b: (-0.25, -0.25))
    grid("lamp", 1, lambda a, b: (-0.4 + 0.8 * a, 1.98, -0.4 + 0.8 * b), lambda a, b: (-0.25, -0.25))
    lines = ["mtllib room.mtl"] + ["v %.6f %.6f %.6f" % p for p in v] + ["vt %.6f %.6f" % t for t in vt]
    for name, faces in f.items():
        lines.append(f"usemtl {name}")
        lines += [f"f {a}/{a} {b}/{b} {c}/{c}" for a, b, c in faces]
    (d / "hall.obj").write_text("\n".join(lines) + "\n")
    return d / "hall.obj"


@pytest.fixture(scope="session")
def textured_hall(native_build, tmp_path_factory):
    from rodent_amd import scene as S
    d = tmp_path_factory.mktemp("hall")
    return S.convert(write_textured_hall(d), d / "hall.rscene")


def write_materials_scene(d):
    """A closed room whose walls exercise every BSDF of the MTL mapping (converter.cpp:858-920): diffuse, Phong only
    (Kd 0), diffuse + Phong mix, mirror (illum 5), glass (illum 7, Ni 1.5, Tf), black (Kd = Ks = 0), and an emitter."""
    d.mkdir(parents=True, exist_ok=True)
    (d / "mats.mtl").write_text(
        "newmtl diffuse\nKd 0.7 0.7 0.6\n"
        "newmtl phong\nKd 0 0 0\nKs 0.8 0.8 0.8\nNs 60\n"
        "newmtl mix\nKd 0.5 0.2 0.2\nKs 0.4 0.4 0.4\nNs 20\n"
        "newmtl mirror\nKs 0.9 0.9 0.9\nillum 5\n"
        "newmtl glass\nKs 1 1 1\nTf 0.9 0.95 0.9\nNi 1.5\nillum 7\n"
        "newmtl black\nKd 0 0 0\nKs 0 0 0\n"
        "newmtl lamp\nKd 0 0 0\nKe 15 15 13\n")
    quads = {                                    # name: four corners, counter-clockwise seen from inside the room
        "diffuse": [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)],              # floor
        "phong":   [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)],            # back
        "mix":     [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)],            # left
        "mirror":  [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)],                # right
        "black":   [(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)],              # ceiling
        "lamp":    [(-0.4, 1.99, -0.4), (0.4, 1.99, -0.4), (0.4, 1.99, 0.4), (-0.4, 1.99, 0.4)],
    }
    lines, nv = ["mtllib mats.mtl"], 0
    for name, q in quads.items():
        for v in q:
            lines.append("v %g %g %g" % v)
        lines += [f"usemtl {name}", f"f {nv + 1} {nv + 2} {nv + 3}", f"f {nv + 1} {nv + 3} {nv + 4}"]
        nv += 4
    # a glass slab standing in the room (two-sided box: six quads)
    x0, x1, y0, y1, z0, z1 = -0.5, 0.2, 0.0, 1.1, -0.2, 0.1
    c = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    for v in c:
        lines.append("v %g %g %g" % v)
    lines.append("usemtl glass")
    for a, b, cc, dd in ((0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)):
        lines += [f"f {nv + a + 1} {nv + b + 1} {nv + cc + 1}", f"f {nv + a + 1} {nv + cc + 1} {nv + dd + 1}"]
    (d / "mats.obj").write_text("\n".join(lines) + "\n")
    return d / "mats.obj"


@pytest.fixture(scope="session")
def materials_scene(native_build, tmp_path_factory):
    from rodent_amd import scene as S
    d = tmp_path_factory.mktemp("materials")
    return S.convert(write_materials_scene(d), d / "mats.rscene")
