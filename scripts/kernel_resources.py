#!/usr/bin/env python
"""Register / scratch / LDS usage of every kernel of one HIP source (lab build), from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python scripts/kernel_resources.py [traversal.hip|render.hip] [--lab] [--grep PATTERN]"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from rodent_amd import build as B
src = next((a for a in sys.argv[1:] if a.endswith(".hip")), "traversal.hip")
pat = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else ""
cmd = [B.HIPCC, *B.HIP_FLAGS, '-DRODENT_HIP_SOURCE_DIGEST="x"', *(["-DRODENT_HIP_LAB"] if "--lab" in sys.argv else []),
    *B.HIP_SOURCE_FLAGS.get(src, []),
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", str(B.CSRC / src), "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy "
        r"\[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = re.sub(r"\(.*", "", demangle(r["name"]).replace("(anonymous namespace)::", "").replace("void ", "", 1))
    if pat and not re.search(pat, name):
        continue
    print(f"{name:90s} vgpr {r.get('VGPRs'):>3s} sgpr {r.get('TotalSGPRs'):>3s} scratch {r.get('ScratchSize'):>4s} vspill "
        f"{r.get('VGPRs Spill'):>2s} sspill {r.get('SGPRs Spill'):>2s} lds {r.get('LDS Size'):>6s} occ {r.get('Occupancy')}")
