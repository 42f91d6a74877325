"""Generates the reference-BUILT BVH fixtures: .bvh files whose hierarchy was made by the REFERENCE's own OBJ loader and
SBVH builder (/root/reference/src/driver/{obj.cpp,bvh.h,...}, compiled from where they lie into oracle/_ref/ref_bvh_builder
by oracle/Makefile.ref).  Runs only in the build container (the reference is not on the GPU box); the outputs are data:

  tests/golden/cornell-refbuilt.bvh            testing/cornell_box.obj (36 triangles): BVH8 + BVH4 + BVH2 blocks
  tests/golden/atrium-decimated-refbuilt.bvh.gz  every 128th face of the procedural atrium (scene_gen atrium, seed 1)

Run from the repo root:  python tests/golden/make_refbuilt.py
The tests rebuild the same OBJ inputs with the in-tree tools (decimate_obj below is imported by them) and compare."""
import gzip
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
G = ROOT / "tests" / "golden"
DECIMATION = 128


def decimate_obj(src: Path, dst: Path, keep_every: int = DECIMATION):
    """Keeps every `keep_every`-th face of an OBJ (all vertices, materials and groups stay)."""
    k = 0
    with open(src) as f, open(dst, "w") as out:
        for line in f:
            if line.startswith("f "):
                if k % keep_every == 0:
                    out.write(line)
                k += 1
            else:
                out.write(line)
    return dst


def atrium_obj(tmp: Path) -> Path:
    from rodent_amd import build
    obj = tmp / "atrium.obj"
    if not obj.exists():
        subprocess.run([str(build.BIN_DIR / "scene_gen"), "atrium", str(obj), "1"], check=True, stdout=subprocess.DEVNULL)
    return decimate_obj(obj, tmp / "atrium-decimated.obj")


def main():
    from rodent_amd import build
    build.build_host(); build.build_reference_tools()
    tool = ROOT / "oracle" / "_ref" / "ref_bvh_builder"
    assert tool.exists(), "oracle/_ref/ref_bvh_builder not built (needs /root/reference)"
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        cmd = [str(tool), str(G / "cornell_box.obj"), str(G / "cornell-refbuilt.bvh")]
        r = subprocess.run(cmd, check=True, capture_output=True, text=True)
        print(r.stdout)
        dec = atrium_obj(d)
        print(subprocess.run([str(tool), str(dec), str(d / "a.bvh")], check=True, capture_output=True, text=True).stdout)
        with open(d / "a.bvh", "rb") as f, gzip.GzipFile(G / "atrium-decimated-refbuilt.bvh.gz", "wb", mtime=0) as z:
            z.write(f.read())
    for p in ("cornell-refbuilt.bvh", "atrium-decimated-refbuilt.bvh.gz"):
        print(p, (G / p).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
