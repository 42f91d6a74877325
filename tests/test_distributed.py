"""Multi-process (gloo, world_size 2, CPU) tests of the sharding + single-gather logic used on N GPUs.
The per-rank compute is done by the CPU oracle here (test infrastructure); on the GPU box the same
partition functions feed the HIP entry points (bench.py, rodent_hip_render_rows)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from rodent_amd import parallel

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from rodent_amd import parallel, scene as S, formats as F
from oracle import binding as O
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
sc = S.Scene(sys.argv[2])
W, H = 96, 50
cam = S.camera_settings((0, 1, 2.7), (0, 0, -1), (0, 1, 0), 60, W, H)
# frame: each rank renders its row band, one gather assembles the film
y0, y1 = parallel.row_band(H, rank, world)
film = np.zeros((H, W, 3), "<f4")
O.render(sc, cam, 2, 2, 8, W, H, film, rows=(y0, y1), threads=1)
full = parallel.gather_film(film[y0:y1], H, dist)
# traversal: contiguous ray ranges, one gather of Hit1
rays = F.read_rays(sys.argv[3], 0.0, 1.0)
a, b = parallel.ray_range(len(rays), rank, world)
hits, _ = O.traverse(2, sc.nodes, sc.tris, rays[a:b])
allhits = parallel.gather_hits(hits, len(rays), dist)
# the tensor forms the GPU path uses (device film / device Hit1 array, completed in place on the root), with uneven
# shares: 7 rows, 1001 rays; and a root other than 0
for root in (0, world - 1):
    t = torch.arange(7 * 5 * 3, dtype=torch.float32).reshape(7, 5, 3)
    y0t, y1t = parallel.row_band(7, rank, world)
    mine = torch.full_like(t, -1.0); mine[y0t:y1t] = t[y0t:y1t]
    got = parallel.gather_film_to_root(mine, dist, root)
    assert (got is None) == (rank != root) and (got is None or torch.equal(got, t))
    # interleaved tiles of 2 rows: 4 tiles, the last of one row
    mine = torch.full_like(t, -1.0)
    for a_, b_ in parallel.row_tiles(7, rank, world, 2): mine[a_:b_] = t[a_:b_]
    got = parallel.gather_film_to_root(mine, dist, root, tile_rows=2)
    assert (got is None) == (rank != root) and (got is None or torch.equal(got, t))
    hb = torch.arange(1001 * 16, dtype=torch.int64).to(torch.uint8)
    at, bt = parallel.ray_range(1001, rank, world)
    mineh = torch.zeros_like(hb); mineh[at * 16: bt * 16] = hb[at * 16: bt * 16]
    goth = parallel.gather_hits_to_root(mineh, 1001, dist, root)
    assert (goth is None) == (rank != root) and (goth is None or torch.equal(goth, hb))
if rank == 0:
    ref, _ = O.render(sc, cam, 2, 2, 8, W, H, threads=1)
    refh, _ = O.traverse(2, sc.nodes, sc.tris, rays)
    assert full is not None and allhits is not None
    assert np.array_equal(full, ref), "band films do not reproduce the frame"
    assert allhits.tobytes() == refh.tobytes(), "gathered hits differ"
    print("DIST_OK", world)
else:
    assert full is None and allhits is None, "only the root receives the gathered film / hits"
dist.barrier(); dist.destroy_process_group()
'''


def test_partitions_cover_exactly():
    for n, w in ((2160, 8), (720, 7), (5, 8), (1 << 20, 3)):
        bands = [parallel.row_band(n, r, w) for r in range(w)]
        assert bands[0][0] == 0 and bands[-1][1] == n and all(bands[i][1] == bands[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in bands) - min(b - a for a, b in bands) <= 1
        assert [parallel.ray_range(n, r, w) for r in range(w)] == bands
    assert parallel.row_band(2160, 3, 8) == (810, 1080)                       # cfg5: 270 rows per GPU
    for n, w, rows in ((2160, 8, 16), (144, 3, 10), (50, 2, 16), (7, 8, 16)):  # interleaved tiles: every row exactly once
        owner = np.full(n, -1)
        for r in range(w):
            for a, b in parallel.row_tiles(n, r, w, rows):
                assert (owner[a:b] == -1).all() and 0 <= a < b <= n
                owner[a:b] = r
        assert (owner >= 0).all() and all(owner[y] == (y // rows) % w for y in range(n))
    assert sum(b - a for a, b in parallel.row_tiles(2160, 0, 8)) == 17 * 16 and sum(b - a for a,
        b in parallel.row_tiles(2160, 7, 8)) == 16 * 16


def test_cpp_hosts_partition_like_the_python_hosts(native_build):
    """host/partition.h (what `rodent --ngpu K` / `bench_traversal -ngpu K` shard by) against parallel.row_band / ray_range."""
    tool = native_build.BIN_DIR / "partition_check"
    for n, w in ((2160, 8), (720, 7), (5, 8), (1 << 20, 3), (1001, 2), (0, 4), (64, 1)):
        out = subprocess.run([str(tool), str(n), str(w)], check=True, capture_output=True, text=True).stdout.split()
        got = [(int(out[3 * r + 1]), int(out[3 * r + 2])) for r in range(w)]
        assert got == [parallel.row_band(n, r, w) for r in range(w)] == [parallel.ray_range(n, r, w) for r in range(w)], (n, w)
    for n, w, rows in ((2160, 8, 16), (144, 3, 10), (50, 2, 16), (7, 8, 16), (0, 2, 16)):            # interleaved row tiles
        out = subprocess.run([str(tool), str(n), str(w), str(rows)], check=True, capture_output=True, text=True).stdout.split()
        got = [(int(out[k]), int(out[k + 1]), int(out[k + 2])) for k in range(0, len(out), 3)]
        assert got == [(r, a, b) for r in range(w) for a, b in parallel.row_tiles(n, r, w, rows)], (n, w, rows)


def test_gather_transport_falls_back_when_rccl_does_not_come_up(native_build):
    """the first multi-GPU run must not fail for want of RCCL.  The decision DeviceGroup::init takes (host/partition.h
    rccl_unused_reason; the GPU suite runs the fallback gather itself: test_bench_traversal_cli) with an injected failure, a real
    ncclCommInitAll error, ranks sharing devices, and the normal case."""
    tool = native_build.BIN_DIR / "partition_check"
    ask = lambda *a: subprocess.run([str(tool), "transport", *[str(x) for x in a]], check=True, capture_output=True,
        text=True).stdout.strip()
    # (one rank: nothing to gather, nothing to fall back from)
    assert ask(8, 8, 0, 0) == "rccl" and ask(2, 8, 0, 0) == "rccl" and ask(1, 1, 0, 1) == "rccl"
    assert ask(8, 8, 0, 1) == "peer copies: RODENT_FORCE_RCCL_INIT_FAILURE"
    assert ask(8, 8, 0, 0, "unhandled system error") == "peer copies: ncclCommInitAll: unhandled system error"
    assert ask(3, 1, 1, 0) == "peer copies: RODENT_SHARE_GPUS: 3 ranks on 1 device(s)" and ask(2, 8, 1, 0) == "rccl"


@pytest.mark.parametrize("world", [2])
def test_two_rank_gloo_bands_and_hits(native_build, tmp_path, world):
    from rodent_amd import scene as S
    S.convert(ROOT / "tests/golden/cornell_box.obj", tmp_path / "c.rscene")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(ROOT), str(tmp_path / "c.rscene"),
        str(ROOT / "tests/golden/cornell-random-4096.rays")],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                  text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert f"DIST_OK {world}" in outs[0]
