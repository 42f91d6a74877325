"""In-tree native build for rodent_amd (no cmake: hipcc / g++ driven directly).

Artefacts (all git-ignored, they travel to the GPU box with the snapshot):
  rodent_amd/lib/librodent_hip.so   HIP kernels + C ABI (include/*.h): the product, default mappings only
  rodent_amd/lib/librodent_hip_lab.so  the same + every measured-and-lost kernel variant and the instrumented
                                    builds (-DRODENT_HIP_LAB); only built when RODENT_HIP_LAB=1 is set
  rodent_amd/bin/<tool>             host tools: bvh_extractor, ray_gen, scene_gen,
                                    fbuf2png, bench_traversal, rodent
  oracle/liboracle.so               CPU parity oracle (test infrastructure only)

`python -m rodent_amd.build` builds everything; `build_all()` is what
__graft_entry__.build() calls.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "rodent_amd"
CSRC = PKG / "csrc"
HOST = PKG / "host"
LIB_DIR = PKG / "lib"
BIN_DIR = PKG / "bin"
OBJ_DIR = ROOT / "build"
ORACLE = ROOT / "oracle"

ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = os.environ.get("HIPCC", str(ROCM / "bin" / "hipcc"))
CXX = os.environ.get("CXX", "g++")
CC = os.environ.get("CC", "gcc")

# -ffp-contract=off: fused multiply-adds are written explicitly (fmaf) in both the
# kernels and the oracle so the two perform the same correctly-rounded operations.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
             "-Wall", "-Wno-unused-function", f"-I{ROOT / 'include'}"]
HOST_FLAGS = ["-O2", "-std=c++17", "-Wall", "-fPIC", f"-I{ROOT / 'include'}"]
ORACLE_FLAGS = ["-O2", "-std=c11", "-Wall", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma"]

HIP_SOURCES = ["traversal.hip", "render.hip", "services.hip"]
# Per-source options.  render.hip: the AMDGPU backend's "max-ilp" instruction scheduling strategy -- the renderer's traversal kernels are
# compiled under an 8-waves-per-SIMD register budget, under which the default strategy serialises their loads; measured on one MI355X
# (profiles/r04_flags_experiment.txt): config 5's frame +3.0 %, config 4 (megakernel) +1.1 %.  Scheduling only: the arithmetic is untouched
# (parity tests).  traversal.hip keeps the default: the benchmark kernel's chunk loop loses 2 % under max-ilp.
HIP_SOURCE_FLAGS = {"render.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
HIP_LIB_HOST_SOURCES = ["image.cpp"]             # host code the library links: texture decoders of rodent_load_png / _jpg
HOST_LIB_SOURCES = ["mesh.cpp", "bvh_build.cpp", "atrium.cpp", "stress_scenes.cpp", "scene.cpp", "image.cpp"]
HOST_TOOLS = ["bvh_extractor", "ray_gen", "scene_gen", "fbuf2png", "converter", "tex_dump", "buffer_tool", "partition_check"]
HIP_TOOLS = {"bench_traversal": [], "rodent": ["mesh.o", "bvh_build.o", "scene.o", "image.o"]}   # tool -> host objects it links


def _newer(target: Path, *deps: Path) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def _run(cmd, **kw):
    print("+", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, **kw)


def _headers():
    return (list((ROOT / "include").glob("*.h")) + list(HOST.glob("*.h")) + list(CSRC.glob("*.h")) + list(CSRC.glob("*.hpp"))
            + list((CSRC / "lab").glob("*.h")) + list((CSRC / "lab").glob("*.inc")))          # csrc/lab: compiled into the lab build only


def _hip_lib_inputs():
    return [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()] + [HOST / s for s in HIP_LIB_HOST_SOURCES]


def source_digest() -> str:
    """sha256 (16 hex digits) over everything librodent_hip.so is compiled from: its sources and every header they can include.
    The build passes it to the compiler (-DRODENT_HIP_SOURCE_DIGEST) and the library returns it from rodent_hip_source_digest():
    whoever loads a prebuilt .so can check that it was built from the sources lying next to it (tests, bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_hip_lib_inputs() + _headers(), key=lambda p: str(p.relative_to(ROOT))):
        h.update(str(f.relative_to(ROOT)).encode())
        h.update(f.read_bytes())
    # ... and every option that can change the code (everything but the include path, which differs between checkouts of the same sources)
    h.update(repr(([f for f in HIP_FLAGS if not f.startswith("-I")], sorted(HIP_SOURCE_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def build_hip_lib(force: bool = False) -> Path:
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    out = LIB_DIR / "librodent_hip.so"
    srcs = _hip_lib_inputs()
    want = source_digest()
    digest = f'-DRODENT_HIP_SOURCE_DIGEST="{want}"'

    def stale(lib):
        # the digest the library was compiled with is also kept in a file beside it: whether a prebuilt .so belongs to these sources can
        # be decided WITHOUT loading it (a process that has dlopen()ed the old file keeps the old mapping whatever is rebuilt afterwards)
        side = lib.with_suffix(".so.digest")
        return force or _newer(lib, *srcs, *_headers()) or not side.exists() or side.read_text().strip() != want

    def compile_to(lib, *extra):
        # one object per source (their options differ), compiled side by side, then one link
        OBJ_DIR.mkdir(parents=True, exist_ok=True)
        objs, procs = [], []
        for src in srcs:
            obj = OBJ_DIR / f"{lib.stem}.{src.stem}.o"
            cmd = [HIPCC, *HIP_FLAGS, digest, *extra, *HIP_SOURCE_FLAGS.get(src.name, []), "-c", src, "-o", obj]
            print("+", " ".join(str(c) for c in cmd), flush=True)
            procs.append((subprocess.Popen([str(c) for c in cmd]), cmd)); objs.append(obj)
        for proc, cmd in procs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, [str(c) for c in cmd])
        _run([HIPCC, HIP_FLAGS[0], "-shared", "-fPIC", *objs, "-lz", "-o", lib])
        lib.with_suffix(".so.digest").write_text(want + "\n")
    if stale(out):
        compile_to(out)
    if os.environ.get("RODENT_HIP_LAB", "0") not in ("", "0"):
        lab = LIB_DIR / "librodent_hip_lab.so"
        if stale(lab):
            compile_to(lab, "-DRODENT_HIP_LAB")
    return out


def build_host(force: bool = False) -> list[Path]:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    BIN_DIR.mkdir(parents=True, exist_ok=True)
    objs = []
    for s in HOST_LIB_SOURCES:
        o = OBJ_DIR / (Path(s).stem + ".o")
        if force or _newer(o, HOST / s, *_headers()):
            _run([CXX, *HOST_FLAGS, "-c", HOST / s, "-o", o])
        objs.append(o)
    outs = []
    for t in HOST_TOOLS:
        src = HOST / "tools" / f"{t}.cpp"
        if not src.exists():
            continue
        out = BIN_DIR / t
        if force or _newer(out, src, *objs, *_headers()):
            _run([CXX, *HOST_FLAGS, src, *objs, "-pthread", "-lz", "-o", out])
        outs.append(out)
    return outs


def build_hip_tools(force: bool = False) -> list[Path]:
    """Host CLIs that link the HIP library (bench_traversal, rodent)."""
    lib = build_hip_lib(force)
    build_host(force)
    outs = []
    for t, needs in HIP_TOOLS.items():
        src = HOST / "tools" / f"{t}.cpp"
        if not src.exists():
            continue
        out = BIN_DIR / t
        objs = [OBJ_DIR / o for o in needs]
        if force or _newer(out, src, lib, *objs, *_headers()):
            # plain g++: host code only uses the HIP runtime API (hipcc mistakes .o inputs for sources)
            _run([CXX, "-O2", "-std=c++17", "-Wall", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__",
                  f"-I{ROOT / 'include'}", f"-I{ROCM / 'include'}", src, *objs,
                  f"-L{LIB_DIR}", "-lrodent_hip", f"-L{ROCM / 'lib'}", "-lamdhip64", "-lrccl",
                  "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{ROCM / 'lib'}", "-pthread", "-lz", "-o", out])
        outs.append(out)
    return outs


def build_ubench(force: bool = False):
    """Lab microbenchmarks (scripts/ubench/*.hip -> rodent_amd/bin/): only with RODENT_HIP_LAB=1."""
    if os.environ.get("RODENT_HIP_LAB", "0") in ("", "0"):
        return
    BIN_DIR.mkdir(parents=True, exist_ok=True)
    for src in sorted((ROOT / "scripts" / "ubench").glob("*.hip")):
        out = BIN_DIR / src.stem
        if force or _newer(out, src):
            _run([HIPCC, "--offload-arch=gfx950", "-O3", src, "-o", out])


def build_oracle(force: bool = False) -> Path:
    out = ORACLE / "liboracle.so"
    srcs = sorted(ORACLE.glob("*.c"))
    if force or _newer(out, *srcs):
        _run([CC, *ORACLE_FLAGS, "-shared", *srcs, "-lm", "-o", out])
    cpp = sorted(ORACLE.glob("*_baseline.cpp"))
    if cpp:
        out2 = ORACLE / "libcpu_baseline.so"
        if force or _newer(out2, *cpp, *sorted(ORACLE.glob("*.inc")), out):
            # links the parity oracle for its shading functions (cpu_wavefront.inc); both are test infrastructure
            _run([CXX, "-O3", "-std=c++17", "-march=x86-64-v3", "-fPIC", "-shared", "-pthread", *cpp, f"-L{ORACLE}", "-l:liboracle.so",
                "-Wl,-rpath,$ORIGIN", "-o", out2])
    return out


def build_reference_tools(force: bool = False):
    """oracle/_ref: the reference's own C++ pieces that compile here (optional)."""
    mk = ORACLE / "Makefile.ref"
    if mk.exists() and Path("/root/reference").exists():
        _run(["make", "-s", "-f", mk, "-C", ORACLE])


def build_all(force: bool = False):
    build_host(force)
    build_hip_lib(force)
    build_hip_tools(force)
    build_ubench(force)
    build_oracle(force)
    build_reference_tools(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
