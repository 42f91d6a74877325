#!/usr/bin/env python
"""Frames across the GPUs of one node (SURVEY 8e / BASELINE config 5): one process per GPU, every rank holds the whole
scene and renders its band of image rows with `rodent_hip_render_rows`, one RCCL gather to rank 0 completes the film
(the C++ host does the same without Python: `rodent --ngpu N`).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
      scripts/render_multi_gpu.py [--scene tests/golden/cornell_box.obj] [--width 3840 --height 2160 --spp 64
      --max-path-len 8 --bench 10 --target amdgpu-streaming|amdgpu-megakernel] [-o image.png]

Prints the reference driver's line (driver.cpp:344-347) with whole-job Msamples/s (max over ranks per frame).
With one process it is the single-GPU renderer."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=str(ROOT / "tests" / "golden" / "cornell_box.obj"))
    ap.add_argument("--width", type=int, default=3840); ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=64); ap.add_argument("--max-path-len", type=int, default=8)
    ap.add_argument("--bench", type=int, default=10)
    ap.add_argument("--eye", type=float, nargs=3, default=[0, 1, 2.7]); ap.add_argument("--dir", type=float, nargs=3, default=[0, 0, -1])
    ap.add_argument("--up", type=float, nargs=3, default=[0, 1, 0]); ap.add_argument("--fov", type=float, default=60.0)
    ap.add_argument("--target", default="amdgpu-streaming", choices=["amdgpu-streaming", "amdgpu-megakernel"])
    ap.add_argument("-o", "--output", default="")
    a = ap.parse_args()

    import torch
    from rodent_amd import parallel, render as R, scene as S
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    if not torch.cuda.is_available():
        raise SystemExit("render_multi_gpu.py needs a GPU (no CPU fallback)")
    path = Path(a.scene)
    if path.suffix == ".rscene":
        scene = S.Scene(path)
    else:
        out = Path("/tmp") / f"{path.stem}.rank{rank}.rscene"
        scene = S.convert(path, out)
    cam = S.camera_settings(a.eye, a.dir, a.up, a.fov, a.width, a.height)
    r = R.Renderer(scene, a.width, a.height, a.spp, a.max_path_len, dev=local,
        mapping={"amdgpu-streaming": "streaming", "amdgpu-megakernel": "megakernel"}[a.target])
    y0, y1 = parallel.row_band(a.height, rank, world)
    rates = []
    for it in range(a.bench):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.render_rows(cam, it, y0, y1)                     # synchronous: returns when the band is in the device film
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local}")
        if dist is not None:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        rates.append(a.spp * a.width * a.height / float(dt[0]) / 1e6)
    # the one collective of the path, on the DEVICE film (no host bounce): a view of the library's film memory; rank 0 receives
    # every peer's rows straight into its own film (grouped send / receive), then copies the frame to the host once
    film = parallel.gather_film_to_root(parallel.device_film(local), dist)
    film = film.cpu().numpy() if rank == 0 else None
    r.close()
    if rank == 0:
        if a.output:
            from PIL import Image
            Image.fromarray(R.tonemap(film, a.bench)).save(a.output)
            print(f"Image saved to '{a.output}'")
        rates.sort()
        print(f"# {rates[0]:g}/{rates[len(rates) // 2]:g}/{rates[-1]:g} (min/med/max Msamples/s)  [{world} GPU(s), rows {a.height} -> bands "
            f"of ~{a.height // world}]")
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
