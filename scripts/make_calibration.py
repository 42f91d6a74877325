#!/usr/bin/env python
"""profiles/rNN_calibration.json from the microbenchmark outputs (rodent_amd/bin/valu_peak, vmem_peak run on the GPU box):
the measured peaks bench.py prices the traversal kernel against.
usage: python scripts/make_calibration.py r02"""
import json, re, sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = Path(__file__).resolve().parents[1] / "profiles"
valu = {}
for line in (P / f"{tag}_ubench_valu_peak.txt").read_text().splitlines():
    m = re.match(r"(.+?)\s+waves/SIMD (\d+):\s+[\d.]+ ms\s+([\d.]+) wave-instr/us/SIMD", line)
    if m and m.group(2) == "8":
        valu[m.group(1).strip()] = float(m.group(3))
fetch = {}
for line in (P / f"{tag}_ubench_vmem_peak.txt").read_text().splitlines():
    m = re.match(r"(.+?)\s{2,}(lane-per-node|4-lanes-per-node)\s+lanes\s+(\d+)%:\s+[\d.]+ ms\s+([\d.]+) node fetches/ns\s+([\d.]+) load "
        r"instr/us/CU", line)
    if m and m.group(2) == "lane-per-node":
        fetch.setdefault(m.group(1).strip(), {})[m.group(3)] = {"node_fetches_per_ns": float(m.group(4)),
            "load_instr_per_us_per_cu": float(m.group(5))}
out = {
    "source": [f"profiles/{tag}_ubench_valu_peak.txt", f"profiles/{tag}_ubench_vmem_peak.txt"],
    "valu_issue_wave_instr_per_us_per_simd_8_waves": valu,
    "valu_issue_peak": valu["mix fma/min/max/cndmask/cmp"],          # the instruction mix of the traversal loop
    "node_fetch": fetch,
    "node_fetch_peak_coherent": fetch["coherent (16 lanes share a node), 2 MiB set"]["100"]["node_fetches_per_ns"],
    "node_fetch_peak_scattered_l2": fetch["scattered, 2 MiB set (L2)"]["100"]["node_fetches_per_ns"],
    "node_fetch_peak_scattered_mall": fetch["scattered, 64 MiB set (MALL)"]["100"]["node_fetches_per_ns"],
    "simds": 1024, "cus": 256,
}
(P / f"{tag}_calibration.json").write_text(json.dumps(out, indent=1))
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))
