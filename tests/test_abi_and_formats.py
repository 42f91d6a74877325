"""CPU tests: the C-ABI library loads and exports what include/*.h declares; file
formats round-trip; the host tools speak the reference's formats and CLIs."""
import ctypes
import re
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from rodent_amd import formats as F

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    names = []
    for h in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names += re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(native_build):
    from rodent_amd import abi
    lib = ctypes.CDLL(str(abi.LIB_PATH))
    decl = declared_functions()
    assert "amdgpu_intersect_single_ray1_bvh2_tri1" in decl and "hip_traverse_bvh8_tri4_async" in decl
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(abi.EXPORTS) <= set(decl)


def test_introspection_without_gpu(native_build):
    from rodent_amd import abi
    assert abi.variants(2)[:2] == ["top", "fast"] and abi.variants(4)[:2] == ["top", "single"] and abi.variants(8)[:2] == ["top",
        "single"] and abi.variants(3) == []
    assert "k_bvh2_top_auto" in abi.kernel_name(2, 0) and "k_bvh2_single" in abi.kernel_name(2,
        1) and "k_wide_top_persist<true,8" in abi.kernel_name(8, 0, any_hit=True) and "k_wide_single<false,4" in abi.kernel_name(4, 1)
    assert abi.lib().rodent_hip_device_count() >= 0
    # the product library ships the default mappings only: the measured-and-lost kernels and the instrumented builds
    # are in the lab build (RODENT_HIP_LAB=1)
    if not abi.LAB:
        assert abi.lib().rodent_hip_is_lab_build() == 0
        assert abi.variants(2) == ["top", "fast", "fast-noxcd", "phased", "sorted",
            "refill"] and not any(n.startswith(("stats-", "trace-")) for w in (2, 4, 8) for n in abi.variants(w))


def test_no_cpu_fallback(native_build):
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    from rodent_amd import abi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        abi.DeviceBvh.load(ROOT / "tests/golden/cornell.bvh", 2)


def test_product_does_not_import_oracle():
    # build.py compiles the checker (building is not using); no product module may import or load it
    for py in (ROOT / "rodent_amd").rglob("*.py"):
        text = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), py
        if py.name != "build.py":
            assert "oracle" not in text, py
    for src in list((ROOT / "rodent_amd").rglob("*.hip")) + list((ROOT / "rodent_amd").rglob("*.cpp")) + list((ROOT
        / "rodent_amd").rglob("*.h")):
        for line in src.read_text().splitlines():
            if line.strip().startswith("#include"):
                assert "oracle" not in line, src


def test_bvh_roundtrip(tmp_path, cornell):
    out = tmp_path / "rt.bvh"
    blocks = [(F.BVH8_TRI4, *cornell.blocks[8]), (F.BVH4_TRI4, *cornell.blocks[4]), (F.BVH2_TRI1, *cornell.blocks[2])]
    F.write_bvh(out, blocks)
    assert out.read_bytes() == cornell.bvh_path.read_bytes()      # byte-identical to the C++ writer
    for b, n, t in blocks:
        n2, t2 = F.read_bvh(out, b)
        assert n2.tobytes() == n.tobytes() and t2.tobytes() == t.tobytes()


def test_bvh_block_header_layout(cornell):
    data = cornell.bvh_path.read_bytes()
    assert struct.unpack_from("<I", data, 0)[0] == 0x95CBED1F            # load_bvh.h:24
    offset, btype, n_nodes, n_tris = struct.unpack_from("<QIII", data, 4)
    assert btype == 3                                                     # BVH8 first (bvh_extractor.cpp:82-90)
    assert offset == 12 + 256 * n_nodes + 224 * n_tris                    # extract_bvh4_8.cpp:12-14
    nxt = struct.unpack_from("<QI", data, 4 + 8 + offset)
    assert nxt[1] == 2


def test_bvh_errors(tmp_path):
    bad = tmp_path / "bad.bvh"
    bad.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError):
        F.read_bvh(bad, F.BVH2_TRI1)
    only8 = tmp_path / "only8.bvh"
    F.write_bvh(only8, [(F.BVH8_TRI4, np.zeros(1, F.NODE8), np.zeros(1, F.TRI4))])
    with pytest.raises(ValueError):
        F.read_bvh(only8, F.BVH2_TRI1)


def test_rays_roundtrip_and_errors(tmp_path, cornell):
    rays = cornell.ray_sets["random"]
    p = tmp_path / "r.rays"
    F.write_rays(p, rays)
    assert p.stat().st_size == 24 * len(rays)                            # load_rays.h:71
    back = F.read_rays(p, 0.0, 1.0)
    assert back.tobytes() == rays.tobytes()
    (tmp_path / "odd.rays").write_bytes(b"\0" * 25)
    with pytest.raises(ValueError):
        F.read_rays(tmp_path / "odd.rays")
    empty = tmp_path / "empty.rays"
    empty.write_bytes(b"")
    assert len(F.read_rays(empty)) == 0


def test_ray_gen_primary_matches_formula(tmp_path, native_build):
    out = tmp_path / "p.rays"
    subprocess.run([native_build.BIN_DIR / "ray_gen", "primary", "0", "1", "2.7", "0", "0", "-1", "0", "1", "0", "60", "8", "4", out],
        check=True)
    raw = np.fromfile(out, "<f4").reshape(4, 8, 6)
    assert np.all(raw[..., :3] == np.float32([0, 1, 2.7]))
    scale = np.float32(np.tan(60 * (np.pi / 360.0)))
    # ray_gen.cpp:41-52: rows top to bottom, pixel centres, unnormalised
    for row, i in enumerate(range(3, -1, -1)):
        for j in range(8):
            kx = np.float32(2 / 8) * np.float32(j + 0.5) - 1
            ky = np.float32(2 / 4) * np.float32(i + 0.5) - 1
            exp = np.float32([kx * scale, ky * np.float32(4 / 8) * scale, -1.0])
            assert np.allclose(raw[row, j, 3:], exp, rtol=1e-6, atol=1e-7)


def test_ray_gen_random_is_seeded_and_inside_bounds(tmp_path, native_build, cornell):
    a, b = tmp_path / "a.rays", tmp_path / "b.rays"
    for o in (a, b):
        subprocess.run([native_build.BIN_DIR / "ray_gen", "random", cornell.bvh_path, "1000", "7", o], check=True)
    assert a.read_bytes() == b.read_bytes()
    raw = np.fromfile(a, "<f4").reshape(-1, 6)
    n4, _ = cornell.blocks[4]
    lo = n4["bounds"][0][[0, 2, 4]].min(axis=1); hi = n4["bounds"][0][[1, 3, 5]].max(axis=1)
    assert (raw[:, :3] >= lo - 1e-5).all() and (raw[:, :3] <= hi + 1e-5).all()
    end = raw[:, :3] + raw[:, 3:]
    assert (end >= lo - 1e-4).all() and (end <= hi + 1e-4).all()


def test_cli_errors(native_build, cornell):
    bt = native_build.BIN_DIR / "bench_traversal"
    r = subprocess.run([bt], capture_output=True, text=True)
    assert r.returncode == 1 and "No BVH file specified" in r.stderr          # bench_traversal.cpp:220-223
    r = subprocess.run([bt, "-bvh", "x.bvh"], capture_output=True, text=True)
    assert r.returncode == 1 and "No ray file specified" in r.stderr
    r = subprocess.run([bt, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown option" in r.stderr
    r = subprocess.run([bt, "-bvh", "x", "-ray", "y", "-gpu", "cuda"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown GPU platform" in r.stderr
    # the reference's CPU variants are not options of this tool: it says where they live
    for flag in ("-s", "--single", "-p", "--packet"):
        r = subprocess.run([bt, "-bvh", "x", "-ray", "y", flag, "-gpu", "hip"], capture_output=True, text=True)
        assert r.returncode == 1 and "CPU traversal variants" in r.stderr and "oracle/cpu_bench_traversal.py" in r.stderr
    r = subprocess.run([bt, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--single" not in r.stdout and "-ngpu" in r.stdout
    r = subprocess.run([bt, "-bvh", str(cornell.bvh_path), "-ray", "y"], capture_output=True, text=True)
    assert r.returncode == 1 and "disabled at compile-time" in r.stderr       # no CPU path in the product


def test_library_was_built_from_the_sources_next_to_it(native_build):
    """librodent_hip.so travels prebuilt (git-ignored, shipped to the GPU box with the snapshot): it carries the digest of the
    sources it was compiled from (rodent_amd/build.py source_digest -> -DRODENT_HIP_SOURCE_DIGEST -> rodent_hip_source_digest()),
    which must be the digest of the sources in this tree -- a stale binary cannot pass for the code under test."""
    from rodent_amd import abi, build
    assert abi.lib().rodent_hip_source_digest().decode() == build.source_digest() and abi.built_from_these_sources()
    assert len(build.source_digest()) == 16 and build.source_digest() != "unknown"


def test_fbuf2png(tmp_path, native_build):
    from PIL import Image
    t = np.linspace(0, 4, 16 * 8, dtype="<f4")
    t.tofile(tmp_path / "x.fbuf")
    subprocess.run([native_build.BIN_DIR / "fbuf2png", "-n", "-sx", "16", "-sy", "8", tmp_path / "x.fbuf", tmp_path / "x.png"], check=True)
    im = np.array(Image.open(tmp_path / "x.png"))
    assert im.shape == (8, 16, 4) and im[-1, -1, 0] == 255 and im[0, 0, 0] == 0 and (im[..., 3] == 255).all()
    exp = (255.0 * t / t.max()).astype(np.uint8).reshape(8, 16)              # fbuf2png.cpp:108-110
    assert np.array_equal(im[..., 0], exp)


def test_top_image_node_set_is_a_breadth_first_prefix(cornell):
    """rodent_amd.topimage restates which nodes the default traversal kernel keeps in LDS (bench.py's accounting): the root
    first, every node after its parent, no node twice, never more than the capacity, the whole tree when it is small."""
    from rodent_amd import topimage
    nodes, _ = cornell.blocks[2]
    child = np.asarray(nodes["child"]).reshape(-1, 2)
    inner = 1 + int((child > 0).sum())                            # the root + every inner child
    for cap in (1, 3, 15, 255):
        ids = topimage.image_nodes(nodes, cap)
        assert ids[0] == 0 and len(ids) == min(cap, inner) and len(set(ids.tolist())) == len(ids)
        seen = {0}
        for i in ids:
            assert int(i) in seen                                  # reached from a node earlier in the list
            seen.update(int(c) - 1 for c in child[i] if c > 0)


def test_packet_model_reproduces_the_oracle_at_threshold_65():
    """scripts/model_packet.py (the wave-packet traversal that was modelled and not built): with every subtree falling back at
    the root (T = 65) the model IS the per-lane kernel and must reproduce oracle B1 bit for bit (the script asserts it), and no packet mode
    may change a hit record on the Cornell fixtures."""
    import subprocess, sys
    from conftest import ROOT
    for rays, tmax in (("cornell-primary-64x64.rays", "5000"), ("cornell-random-4096.rays", "1")):
        r = subprocess.run([sys.executable, str(ROOT / "scripts/model_packet.py"), "--bvh", str(ROOT / "tests/golden/cornell.bvh"),
            "--rays", str(ROOT / "tests/golden" / rays),
                            "--tmax", tmax, "--thresholds", "8,32", "--buildable"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "hits == B1: yes" in r.stdout
        rows = [l.split("|") for l in r.stdout.splitlines() if l.startswith(("immediate", "deferred", "buildable"))]
        assert len(rows) == 6 and all(row[-1].split() == ["0", "0"] for row in rows), r.stdout
