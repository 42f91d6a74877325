"""Textures: converter tables, texture lookup and per-hit materials of the CPU oracle.

Reference semantics: src/render/image.impala:24-92 (RGBA8 -> colour, repeat border, bilinear filter),
src/driver/converter.cpp:595-610,749-768,875-906 (map_Kd / map_Ks, mix weight from the looked-up colours),
src/driver/image.cpp:10-18,85 (gamma correction and vertical flip at load time)."""
import subprocess

import numpy as np
import pytest
from PIL import Image

from rodent_amd import scene as S


def load_like_reference(path):
    a = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8).copy()
    lut = (np.power(np.arange(256, dtype=np.float32) * np.float32(1 / 255.0), np.float32(2.2)) * np.float32(255.0)).astype(np.uint8)
    a[..., :3] = lut[a[..., :3]]
    return a[::-1]


def test_converter_builds_texture_tables(textured_scene):
    sc, d = textured_scene
    assert len(sc.textures) == 3 and len(sc.texels) == 32 * 32 + 48 * 24 + 32 * 32
    names = ["floor", "back", "side", "lamp"]
    mats = {n: sc.materials[i] for i, n in enumerate(names)}
    assert mats["floor"]["type"] == 1 and mats["floor"]["tex_kd"] == 1 and mats["floor"]["tex_ks"] == 0
    assert mats["back"]["type"] == 3 and mats["back"]["tex_kd"] == 2 and mats["back"]["tex_ks"] == 3
    assert mats["side"]["tex_kd"] == 0 and mats["lamp"]["emissive"] == 1
    t = sc.textures
    assert (t["width"].tolist(), t["height"].tolist(), t["offset"].tolist()) == ([32, 48, 32], [32, 24, 32], [0, 1024, 1024 + 1152])
    # texels = the file decoded like the reference's loaders (PNG / TGA exact)
    for k, name in ((0, "checker.png"), (2, "spec.tga")):
        got = sc.texels[t["offset"][k]: t["offset"][k] + 1024].view(np.uint8).reshape(32, 32, 4)
        assert np.array_equal(got, load_like_reference(d / name))
    assert sc.texcoords.shape == sc.vertices.shape and sc.texcoords[:, 2:].max() == 0 and sc.texcoords[:, 0].max() == 2.5


def test_missing_texture_becomes_black_dummy(native_build, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\nmap_Kd nothere.png\nnewmtl l\nKe 1 1 1\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl a\nf 1/1 2/1 3/1\nusemtl l\nf 1/1 3/1 2/1\n")
    r = subprocess.run([native_build.BIN_DIR / "converter", tmp_path / "m.obj", "-o", tmp_path / "m.rscene"], capture_output=True,
        text=True, check=True)
    assert "Cannot load texture 'nothere.png'" in r.stderr
    sc = S.Scene(tmp_path / "m.rscene")
    assert len(sc.textures) == 1 and sc.textures[0]["width"] == 1 and sc.texels.tolist() == [0]


def bilinear_ref(img, uv):
    """numpy float32 restatement of image.impala:48-86 on an (h, w, 4) uint8 image with row 0 = bottom."""
    f = np.float32
    h, w = img.shape[:2]
    col = img[..., :3].astype(f) * f(1 / 255.0)
    out = np.zeros((len(uv), 3), f)
    for i, (tu, tv) in enumerate(uv.astype(f)):
        ru, rv = f(tu - np.floor(tu)), f(tv - np.floor(tv))
        u, v = f(ru * f(w)), f(rv * f(h))
        iu, iv = int(u), int(v)
        x0, y0 = min(iu, w - 1), min(iv, h - 1)
        x1, y1 = min(x0 + 1, w - 1), min(y0 + 1, h - 1)
        kx, ky = f(u - f(iu)), f(v - f(iv))
        lerp = lambda a, b, k: f(f(f(1) - k) * a) + f(k * b)
        out[i] = lerp(lerp(col[y0, x0], col[y0, x1], kx), lerp(col[y1, x0], col[y1, x1], kx), ky)
    return out


def test_oracle_texture_lookup_is_repeat_bilinear(oracle, textured_scene):
    sc, d = textured_scene
    rng = np.random.default_rng(7)
    uv = np.concatenate([rng.uniform(-3, 3, (400, 2)),
        [[0, 0], [1, 1], [0.999999, 0.5], [-1e-9, 0.25], [2.5, -0.25], [31.5 / 32, 31.5 / 32]]]).astype("<f4")
    for k, name in ((0, "checker.png"), (2, "spec.tga")):
        got = oracle.tex_lookup(sc, k, uv)
        assert np.array_equal(got, bilinear_ref(load_like_reference(d / name), uv))
    # orientation: the red marker sits at the TOP-left of the file = high v, low u
    assert np.allclose(oracle.tex_lookup(sc, 0, np.array([[2 / 32, 30 / 32]], "<f4"))[0], [1.0, 0, 0], atol=1e-6)


def test_per_hit_material_follows_the_textures(oracle, textured_scene):
    sc, _ = textured_scene
    floor_prim, back_prim, side_prim = 0, 2, 4
    assert sc.indices[floor_prim, 3] == 0 and sc.indices[back_prim, 3] == 1 and sc.indices[side_prim, 3] == 2
    # floor: kd = the checker at the interpolated texture coordinate (vertex 0 has vt (0,0), vertex 1 (2.5,0), vertex 2 (2.5,2.5))
    for u, v in ((0.1, 0.2), (0.33, 0.33), (0.7, 0.05)):
        m = oracle.hit_material(sc, floor_prim, u, v)
        tc = (np.float32(1 - u - v) * sc.texcoords[sc.indices[floor_prim, 0], :2] + np.float32(u) * sc.texcoords[sc.indices[floor_prim, 1],
            :2]
              + np.float32(v) * sc.texcoords[sc.indices[floor_prim, 2], :2])
        assert np.allclose(m["kd"], oracle.tex_lookup(sc, 0, tc[None])[0], atol=2e-6)
        assert m["type"] == 1 and m["tex_kd"] == 1
    # back wall: mix weight = lum(ks) / (lum(ks) + lum(kd)) of the looked-up colours; zero where the specular map is black
    ks_seen = set()
    for u in np.linspace(0.02, 0.9, 23):
        m = oracle.hit_material(sc, back_prim, float(u), 0.05)
        lum = lambda c: np.float32(c[0]) * np.float32(0.2126) + np.float32(c[1]) * np.float32(0.7152) + np.float32(c[2]) * np.float32(
            0.0722)
        ls, ld = lum(m["ks"]), lum(m["kd"])
        assert np.isclose(m["mix_k"], 0.0 if ls + ld == 0 else ls / (ls + ld), rtol=1e-6)
        ks_seen.add(round(float(m["ks"][0]), 3))
    assert len(ks_seen) > 2 and min(ks_seen) == 0.0
    # untextured material: the table entry itself
    assert oracle.hit_material(sc, side_prim, 0.3, 0.3).tobytes() == sc.materials[2].tobytes()


def test_textured_render_shows_the_checker(oracle, textured_scene):
    sc, _ = textured_scene
    W, H = 96, 64
    cam = S.camera_settings((0.3, 1.0, 3.2), (-0.1, -0.25, -1), (0, 1, 0), 50, W, H)
    film = None
    for it in range(4):
        film, counts = oracle.render(sc, cam, it, 8, 6, W, H, film)
    img = film / 4
    assert np.isfinite(img).all() and img.mean() > 0.02
    floor = img[44:62, 20:80]                                    # floor region: yellow and blue checker fields
    yellowish = (floor[..., 0] > 1.5 * floor[..., 2]).mean(); bluish = (floor[..., 2] > 1.5 * floor[..., 0]).mean()
    assert yellowish > 0.1 and bluish > 0.08
    plain = S.Scene.__new__(S.Scene); plain.__dict__.update(sc.__dict__)
    plain.materials = sc.materials.copy(); plain.materials["tex_kd"] = 0; plain.materials["tex_ks"] = 0
    film2, _ = oracle.render(plain, cam, 0, 8, 6, W, H)
    f2 = film2[44:62, 20:80]
    assert (f2[..., 2] > 1.5 * f2[..., 0]).mean() < 0.02          # without the map the floor is white
