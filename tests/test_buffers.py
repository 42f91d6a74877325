"""The reference's LZ4 buffer files (src/driver/buffer.h, converter.cpp:403-437): the C++ block codec
written for this repo (lz4.h is not in the image) against liblz4 itself, in both directions, and the converter's
data directory against the .rscene tables."""
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from rodent_amd import formats as F, scene as S


def cases():
    rng = np.random.default_rng(5)
    yield "empty", b""
    yield "one", b"x"
    yield "twelve", b"abcabcabcabc"
    yield "thirteen", b"abcabcabcabca"
    yield "zeros", bytes(1 << 20)
    yield "random", rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    yield "floats", np.repeat(rng.normal(size=5000).astype("<f4"), 3).tobytes()
    yield "long-literals-then-match", rng.integers(0, 256, 60_000, dtype=np.uint8).tobytes() * 2
    # offset beyond 65535: must not be used
    yield "far-match", (rng.integers(0, 256, 66_000, dtype=np.uint8).tobytes() + b"\\x00" * 100) * 2
    yield "runs", b"".join(bytes([k % 251]) * (k % 37 + 1) for k in range(20000))


@pytest.mark.parametrize("name,raw", list(cases()), ids=[n for n, _ in cases()])
def test_block_codec_against_liblz4(native_build, tmp_path, name, raw):
    tool = native_build.BIN_DIR / "buffer_tool"
    (tmp_path / "raw").write_bytes(raw)
    # ours -> liblz4
    subprocess.run([tool, "pack", tmp_path / "raw", tmp_path / "ours.bin"], check=True)
    assert F.read_buffer_file(tmp_path / "ours.bin").tobytes() == raw
    packed = (tmp_path / "ours.bin").read_bytes()
    if name in ("zeros", "floats", "runs", "long-literals-then-match"):
        assert len(packed) < 0.7 * len(raw)                       # it does compress
    # liblz4 -> ours
    F.write_buffer_file(tmp_path / "theirs.bin", np.frombuffer(raw, np.uint8))
    subprocess.run([tool, "unpack", tmp_path / "theirs.bin", tmp_path / "back"], check=True)
    assert (tmp_path / "back").read_bytes() == raw


def test_corrupt_buffers_are_rejected(native_build, tmp_path):
    tool = native_build.BIN_DIR / "buffer_tool"
    F.write_buffer_file(tmp_path / "ok.bin", np.arange(5000, dtype="<i4"))
    good = (tmp_path / "ok.bin").read_bytes()
    for label, bad in (("truncated", good[:-7]), ("wrong size", good[:3] + b"\\x7f" + good[4:]),
        ("bad offset", good[:8] + b"\\x00\\x01\\x00" + good[11:])):
        (tmp_path / "bad.bin").write_bytes(bad)
        r = subprocess.run([tool, "unpack", tmp_path / "bad.bin", tmp_path / "x"], capture_output=True, text=True)
        assert r.returncode != 0 and "Invalid buffer file" in r.stderr, label


def test_converter_writes_and_reads_the_reference_data_directory(native_build, tmp_path):
    data = tmp_path / "data"; data.mkdir()
    conv = native_build.BIN_DIR / "converter"
    subprocess.run([conv, GOLDEN / "cornell_box.obj", "-o", tmp_path / "c.rscene", "--data-dir", data], check=True, capture_output=True)
    sc = S.Scene(tmp_path / "c.rscene")
    for name, arr in (("vertices", sc.vertices), ("normals", sc.normals), ("face_normals", sc.face_normals), ("texcoords", sc.texcoords)):
        assert np.array_equal(F.read_buffer_file(data / f"{name}.bin", "<f4").reshape(-1, 4), arr), name
    assert np.array_equal(F.read_buffer_file(data / "indices.bin", "<i4").reshape(-1, 4), sc.indices)
    assert np.array_equal(F.read_buffer_file(data / "light_ids.bin", "<i4"), sc.light_ids)
    nodes, tris = F.read_bvh_bin(data / "bvh.bin")
    assert nodes.tobytes() == sc.nodes.tobytes() and tris.tobytes() == sc.tris.tobytes()
    assert np.array_equal(F.read_buffer_file(data / "light_areas.bin", "<f4"), sc.lights["inv_area"])
    assert np.array_equal(F.read_buffer_file(data / "light_colors.bin", "<f4").reshape(-1, 4)[:, :3], sc.lights["color"][:, :3])
    # the other way: the same tables packed by liblz4 (a different compressor), with a foreign layout first in bvh.bin
    theirs = tmp_path / "theirs"; theirs.mkdir()
    for f in data.iterdir():
        if f.name != "bvh.bin":
            F.write_buffer_file(theirs / f.name, F.read_buffer_file(f))
    n8, t8 = F.read_bvh(GOLDEN / "cornell.bvh", F.BVH8_TRI4)
    F.write_bvh_bin(theirs / "bvh.bin", n8, t8)                   # a BVH8/Tri4 layout the loader must skip (interface.cpp:450-451)
    F.write_bvh_bin(theirs / "bvh.bin", nodes, tris, append=True)
    r = subprocess.run([conv, GOLDEN / "cornell_box.obj", "-o", tmp_path / "d.rscene", "--verify-data-dir", theirs], capture_output=True,
        text=True, check=True)
    assert "match the converted scene" in r.stdout
    F.write_buffer_file(theirs / "indices.bin", sc.indices[::-1])
    r = subprocess.run([conv, GOLDEN / "cornell_box.obj", "-o", tmp_path / "d.rscene", "--verify-data-dir", theirs], capture_output=True,
        text=True)
    assert r.returncode != 0 and "differ from the converted scene" in r.stderr
