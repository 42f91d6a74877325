#!/bin/bash
# round 5, call AI: what binds k_shade? SQ counters per renderer kernel on config 5's scene (32 spp)
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
C="rodent_amd/bin/rodent --scene data/atrium.obj --bench 1 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 3840 --height 2160 --spp 32 --max-path-len 8"
timeout -k 5 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/r05/shade_sq -o rodent -- $C > gpurun_out/r05/shade_sq.log 2>&1
python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/r05/shade_sq/rodent_counter_collection.csv")):
    acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    n = len(next(iter(c.values())))
    print(k, "calls", n, {name: round(sum(v) / len(v), 1) for name, v in c.items()})
PY
