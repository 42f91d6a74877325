"""Texture image decoders of the scene converter (rodent_amd/host/image.cpp) against PIL's decoders.

The reference links libpng / libjpeg (src/driver/image.cpp); this repo carries its own PNG / JPEG / TGA decoders,
so they are checked against an independent implementation: PNG and TGA bit for bit, JPEG within 3 levels (IDCT and
chroma upsampling differ between implementations).  Expected texels = PIL's RGBA, gamma-corrected and flipped the
way the reference's loaders do (image.cpp:10-18,85)."""
import subprocess

import numpy as np
import pytest
from PIL import Image


def expected(pil_img):
    a = np.asarray(pil_img.convert("RGBA"), dtype=np.uint8).copy()
    lut = (np.power(np.arange(256, dtype=np.float32) * np.float32(1 / 255.0), np.float32(2.2)) * np.float32(255.0)).astype(np.uint8)
    a[..., :3] = lut[a[..., :3]]
    return a[::-1]


def decode(native_build, path, tmp_path):
    out = tmp_path / "out.rgba"
    subprocess.run([native_build.BIN_DIR / "tex_dump", path, out], check=True, capture_output=True)
    raw = out.read_bytes()
    w, h = np.frombuffer(raw[:8], "<i4")
    return np.frombuffer(raw[8:], np.uint8).reshape(h, w, 4)


def picture(w, h, seed=1):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 5) % 256], -1).astype(np.uint8)
    base[h // 4: h // 2, w // 4: w // 2] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 4, 3), dtype=np.uint8)
    return base


@pytest.mark.parametrize("mode,kwargs", [("RGB", {}), ("RGBA", {}), ("L", {}), ("LA", {}), ("P", {}), ("1", {}), ("I;16", {}),
                                         ("RGB", {"interlace": True}), ("P", {"bits": 4}), ("RGB", {"compress_level": 0})])
def test_png_matches_pil(native_build, tmp_path, mode, kwargs):
    rgb = picture(67, 45)
    kwargs = dict(kwargs)
    if mode == "RGBA":
        im = Image.fromarray(np.dstack([rgb, (rgb[..., 0] // 2 + 64).astype(np.uint8)]), "RGBA")
    elif mode == "I;16":
        im = Image.fromarray((rgb[..., 0].astype(np.uint16) * 257), "I;16")
    elif mode == "P":
        im = Image.fromarray(rgb, "RGB").quantize(16 if kwargs.get("bits") == 4 else 200)
    else:
        im = Image.fromarray(rgb, "RGB").convert(mode)
    path = tmp_path / "t.png"
    interlace = kwargs.pop("interlace", False)
    im.save(path, **kwargs)
    if interlace:                                   # PIL cannot write Adam7: re-encode the scanlines by hand
        path = write_adam7(tmp_path / "i.png", rgb)
        im = Image.fromarray(rgb, "RGB")
    got = decode(native_build, path, tmp_path)
    ref = Image.open(path)
    exp = expected(ref if mode != "I;16" else Image.fromarray((np.asarray(ref) >> 8).astype(np.uint8), "L"))
    assert got.shape == exp.shape and np.array_equal(got, exp)


def write_adam7(path, rgb):
    import struct, zlib
    h, w, _ = rgb.shape
    raw = b""
    for xs, ys, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = rgb[ys::dy, xs::dx]
        if sub.size:
            raw += b"".join(b"\x00" + row.tobytes() for row in sub)
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT",
        zlib.compress(raw)) + chunk(b"IEND", b""))
    return path


@pytest.mark.parametrize("subsampling,gray,restart", [(0, False, 0), (2, False, 0), (1, False, 0), (0, True, 0), (2, False, 4)])
def test_jpeg_close_to_pil(native_build, tmp_path, subsampling, gray, restart):
    rgb = picture(83, 61, seed=3)
    rgb[30:, 40:] = np.clip(rgb[30:, 40:].astype(int) // 2 + 60, 0, 255)          # smooth region
    im = Image.fromarray(rgb, "RGB").convert("L" if gray else "RGB")
    path = tmp_path / "t.jpg"
    kw = {"quality": 92, "subsampling": subsampling} if not gray else {"quality": 92}
    if restart:
        kw["restart_marker_blocks"] = restart
    try:
        im.save(path, **kw)
    except TypeError:
        kw.pop("restart_marker_blocks", None); im.save(path, **kw)
    got = decode(native_build, path, tmp_path).astype(int)
    srgb = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)[::-1].astype(int)
    # compare before gamma (undo it approximately is lossy): decode again without gamma through the inverse LUT bound
    lut = (np.power(np.arange(256, dtype=np.float32) * np.float32(1 / 255.0), np.float32(2.2)) * np.float32(255.0)).astype(int)
    lo, hi = lut[np.clip(srgb[..., :3] - 3, 0, 255)], lut[np.clip(srgb[..., :3] + 3, 0, 255)]
    inside = (got[..., :3] >= lo) & (got[..., :3] <= hi)
    frac = inside.mean()
    assert got.shape == srgb.shape and (got[..., 3] == 255).all()
    assert frac > 0.99, frac


@pytest.mark.parametrize("subsampling,gray,restart,quality,size",
    [(0, False, 0, 92, (83, 61)), (2, False, 0, 92, (83, 61)), (1, False, 0, 75, (130, 47)), (0, True, 0, 92, (83, 61)),
                                                                   (2, False, 3, 60, (200, 150)), (2, False, 0, 30, (64, 64))])
def test_progressive_jpeg_close_to_pil(native_build, tmp_path, subsampling, gray, restart, quality, size):
    """SOF2 files (spectral selection + successive approximation: DC first / refinement scans, AC bands with end-of-band runs,
    AC refinement passes, several scans per component) decode like libjpeg's output -- the reference reads its textures with
    libjpeg, which takes progressive files (src/driver/image.cpp:185-238)."""
    w, h = size
    rgb = picture(w, h, seed=11)
    rgb[h // 2:, w // 2:] = np.clip(rgb[h // 2:, w // 2:].astype(int) // 2 + 60, 0, 255)          # smooth region
    im = Image.fromarray(rgb, "RGB").convert("L" if gray else "RGB")
    path = tmp_path / "p.jpg"
    kw = {"quality": quality, "progressive": True}
    if not gray:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    try:
        im.save(path, **kw)
    except TypeError:
        kw.pop("restart_marker_blocks", None); im.save(path, **kw)
    assert b"\xff\xc2" in path.read_bytes()[:2000]                                                # really a progressive frame
    got = decode(native_build, path, tmp_path).astype(int)
    srgb = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)[::-1].astype(int)
    lut = (np.power(np.arange(256, dtype=np.float32) * np.float32(1 / 255.0), np.float32(2.2)) * np.float32(255.0)).astype(int)
    lo, hi = lut[np.clip(srgb[..., :3] - 3, 0, 255)], lut[np.clip(srgb[..., :3] + 3, 0, 255)]
    inside = (got[..., :3] >= lo) & (got[..., :3] <= hi)
    assert got.shape == srgb.shape and (got[..., 3] == 255).all()
    assert inside.mean() > 0.99, inside.mean()


def test_arithmetic_coded_jpeg_is_rejected(native_build, tmp_path):
    Image.fromarray(picture(40, 40), "RGB").save(tmp_path / "p.jpg")
    data = bytearray((tmp_path / "p.jpg").read_bytes())
    # SOF9: extended sequential, arithmetic coding
    i = data.index(b"\xff\xc0"); data[i + 1] = 0xC9
    (tmp_path / "a.jpg").write_bytes(bytes(data))
    r = subprocess.run([native_build.BIN_DIR / "tex_dump", tmp_path / "a.jpg", tmp_path / "o"], capture_output=True, text=True)
    assert r.returncode != 0 and "arithmetic" in r.stderr


@pytest.mark.parametrize("rle,alpha,gray", [(False, False, False), (True, False, False), (True, True, False), (False, False, True)])
def test_tga_matches_pil(native_build, tmp_path, rle, alpha, gray):
    rgb = picture(50, 37, seed=5)
    rgb[10:20] = rgb[10:20, :1]                                     # runs for the RLE packets
    if gray:
        im = Image.fromarray(rgb[..., 0], "L")
    elif alpha:
        im = Image.fromarray(np.dstack([rgb, rgb[..., 1]]), "RGBA")
    else:
        im = Image.fromarray(rgb, "RGB")
    path = tmp_path / "t.tga"
    im.save(path, compression="tga_rle" if rle else None)
    assert np.array_equal(decode(native_build, path, tmp_path), expected(Image.open(path)))


def test_bad_files_fail_loudly(native_build, tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png")
    (tmp_path / "x.bmp").write_bytes(b"BM")
    for name, msg in (("x.png", "not a PNG"), ("x.bmp", "cannot determine"), ("missing.jpg", "cannot read")):
        r = subprocess.run([native_build.BIN_DIR / "tex_dump", tmp_path / name, tmp_path / "o"], capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr


def test_greyscale_jpeg_with_2x2_sampling_factors(native_build, tmp_path):
    """A single-component scan is not interleaved: its MCU is one 8x8 block whatever the frame header's sampling factors say
    (ITU T.81 A.2.2); some encoders write greyscale files with 2x2 factors.  Patching the factors of a greyscale file must
    not change a texel (the entropy-coded data are the same)."""
    rgb = picture(83, 61, seed=5)
    path = tmp_path / "g.jpg"
    Image.fromarray(rgb, "RGB").convert("L").save(path, quality=90)
    plain = decode(native_build, path, tmp_path)
    data = bytearray(path.read_bytes())
    pos = data.find(b"\xff\xc0")                                   # SOF0: len(2) precision(1) height(2) width(2) ncomp(1) [id, HV, Tq]
    assert pos > 0 and data[pos + 9] == 1 and data[pos + 11] == 0x11
    data[pos + 11] = 0x22
    (tmp_path / "g22.jpg").write_bytes(bytes(data))
    patched = decode(native_build, tmp_path / "g22.jpg", tmp_path)
    assert patched.shape == plain.shape and np.array_equal(patched, plain)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "g22.jpg")), np.asarray(Image.open(path)))      # PIL agrees that nothing changed
