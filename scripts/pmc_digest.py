#!/usr/bin/env python
"""Mean per dispatch of every counter in the rocprofv3 --pmc passes <dir>/<prefix>*/ (scripts/profile_pmc.sh), for the
kernels whose name contains one of the given substrings (default: the traversal and renderer kernels).  Also writes
<dir>/<prefix>_counters.json.  The JSON carries the hash of the kernel sources it was taken on (rodent_amd/provenance.py).
usage: python scripts/pmc_digest.py gpurun_out/profiles r02_pmc [k_bvh2 ...]"""
import csv, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rodent_amd import provenance

out, prefix = Path(sys.argv[1]), sys.argv[2]
want = sys.argv[3:] or ["k_bvh2", "k_wide", "k_trace", "k_shade", "k_scatter", "k_bin", "k_mega", "k_generate"]
result = {}
for d in sorted(p for p in out.glob(prefix + "*") if p.is_dir()):
    f = next(iter(sorted(d.rglob("*counter_collection.csv"))), None)
    if not f:
        print(f"[{d.name}] no counter_collection.csv"); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if any(w in k for w in want):
            agg[(k.split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(f"== {d.name}")
    for (k, name), v in sorted(agg.items()):
        print(f"   {k[:58]:58s} {name:40s} n={len(v):3d} {sum(v) / len(v):18.1f}")
        result.setdefault(d.name, {}).setdefault(k, {})[name] = sum(v) / len(v)
# bench.py quotes the counters only while this hash holds
result["_meta"] = provenance.stamp("render" if any(w.startswith(("k_trace", "k_shade", "k_scatter", "k_bin", "k_mega", "k_generate"))
    for w in sys.argv[3:]) else "traversal")
json.dump(result, open(out / f"{prefix}_counters.json", "w"), indent=1)
