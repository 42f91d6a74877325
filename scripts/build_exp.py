#!/usr/bin/env python
"""An experiment build of the HIP library beside the shipped one: python scripts/build_exp.py <name> [-DMACRO ...]  ->
rodent_amd/lib/exp_<name>.so (loaded with RODENT_HIP_LIB=rodent_amd/lib/exp_<name>.so; abi.py skips the source-digest check for such a
library)."""
import subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
out = B.LIB_DIR / f"exp_{name}.so"
digest = f'-DRODENT_HIP_SOURCE_DIGEST="{B.source_digest()}"'
objs, procs = [], []
B.OBJ_DIR.mkdir(parents=True, exist_ok=True)
for src in B._hip_lib_inputs():
    obj = B.OBJ_DIR / f"exp_{name}.{src.stem}.o"
    cmd = [B.HIPCC, *B.HIP_FLAGS, digest, *extra, *B.HIP_SOURCE_FLAGS.get(src.name, []), "-c", src, "-o", obj]
    procs.append(subprocess.Popen([str(c) for c in cmd])); objs.append(obj)
assert all(p.wait() == 0 for p in procs)
subprocess.run([str(c) for c in [B.HIPCC, B.HIP_FLAGS[0], "-shared", "-fPIC", *objs, "-lz", "-o", out]], check=True)
print(out)
