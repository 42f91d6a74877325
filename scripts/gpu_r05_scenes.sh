#!/bin/bash
# round 5: the scene x ray-class matrix (scripts/scene_matrix.py), then L2 hit rates of the same launches (one rocprofv3 --pmc pass, kernel-trace only)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
nproc > $O/scene_matrix.txt
timeout 2400 python scripts/scene_matrix.py --json $O/scene_matrix.json 2>&1 | grep -v amdgpu.ids >> $O/scene_matrix.txt
cat $O/scene_matrix.txt
cd /tmp
rm -rf /tmp/pmc_scenes
timeout 1200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_scenes -o s -- python $GRAFT_REPO_ROOT/scripts/scene_matrix.py --pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/pmc_scenes/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
# one dispatch of the traversal kernel per (scene, ray class), in launch order
by = {}
for r in rows:
    if "k_bvh2_top_auto" not in r["Kernel_Name"]: continue
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = open("gpurun_out/r05/scene_matrix_l2.txt", "w")
cells = [f"{s} {k}" for s in ("atrium", "gallery", "crown", "plant") for k in ("primary-for-ao-hits", "primary", "random", "ao")]
for (d, c), name in zip(sorted(by.items(), key=lambda x: int(x[0])), cells):
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        line = f"{name:28s} L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}  ({c['TCC_HIT_sum']:.0f} hits, {c['TCC_MISS_sum']:.0f} misses)"
        print(line); out.write(line + "\n")
PY
