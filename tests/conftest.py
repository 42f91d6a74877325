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
