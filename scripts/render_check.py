#!/usr/bin/env python
"""GPU renderer vs CPU oracle on the Cornell box (small frame), and vs the reference's golden image."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import scene as S, render as R
from oracle import binding as O
from PIL import Image

root = Path(__file__).resolve().parents[1]
sc = S.convert(root / "tests/golden/cornell_box.obj", "/tmp/cornell.rscene")
W, H, SPP, ITERS = 270, 180, 4, 4
cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
r = R.Renderer(sc, W, H, SPP, 64)
film_o = None
for it in range(ITERS):
    r.render(cam, it)
    film_o, cnt = O.render(sc, cam, it, SPP, 64, W, H, film_o)
film_g = r.film()
print("counters", r.counters(), "oracle rays", cnt)
d = np.abs(film_g - film_o)
rel = d / np.maximum(np.abs(film_o), 1e-3)
print("film: gpu mean %.6f oracle mean %.6f  max abs diff %.3e  max rel diff %.3e  pixels>1e-4 rel: %d of %d" %
      (film_g.mean(), film_o.mean(), d.max(), rel.max(), int((rel > 1e-4).any(axis=2).sum()), W * H))
Image.fromarray(R.tonemap(film_g, ITERS)).save("gpurun_out/cornell_gpu_small.png")
# full-size golden-image check
W, H, SPP, ITERS = 1080, 720, 4, 50
cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
r.close()
r = R.Renderer(sc, W, H, SPP, 64)
t0 = time.time()
for it in range(ITERS):
    r.render(cam, it)
dt = time.time() - t0
img = R.tonemap(r.film(), ITERS)
Image.fromarray(img).save("gpurun_out/cornell_gpu.png")
print("1080x720 x %d spp: %.2f s -> %.1f Msamples/s; counters(last frame) %s" % (SPP * ITERS, dt, SPP * ITERS * W * H / dt / 1e6,
    r.counters()))
