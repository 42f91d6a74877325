#!/bin/bash
# round 6, call I: the scene x ray-class matrix and the renderer's SQ counters (config 5's scene, 32 spp) on the round's final sources
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
O=gpurun_out/r06
nproc > $O/scene_matrix.txt
timeout 2400 python scripts/scene_matrix.py --json $O/scene_matrix.json 2>&1 | grep -v amdgpu.ids >> $O/scene_matrix.txt
tail -20 $O/scene_matrix.txt
C="rodent_amd/bin/rodent --scene data/atrium.obj --bench 1 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 3840 --height 2160 --spp 32 --max-path-len 8"
timeout -k 5 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/render_sq -o rodent -- $C > $O/render_sq.log 2>&1
timeout -k 5 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/render_sq2 -o rodent -- $C > $O/render_sq2.log 2>&1
python - <<'PY' | tee gpurun_out/r06/render_counters.txt
import csv, collections
print("# rocprofv3 --pmc (two passes) -- rodent --scene data/atrium.obj --bench 1 ... --width 3840 --height 2160 --spp 32 --max-path-len 8: mean per call of each renderer kernel")
for d in ("render_sq", "render_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"gpurun_out/r06/{d}/rodent_counter_collection.csv")):
        acc[r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        n = len(next(iter(c.values())))
        m = {name: sum(v) / len(v) for name, v in c.items()}
        line = f"{k:28s} calls {n:4d}"
        if "SQ_INSTS_VALU" in m:
            line += (f"  waves {m['SQ_WAVES']:.0f}  VALU/call {m['SQ_INSTS_VALU']:.4g}  SALU/call {m['SQ_INSTS_SALU']:.4g}  lane util {m['SQ_THREAD_CYCLES_VALU'] / 64 / max(m['SQ_ACTIVE_INST_VALU'], 1):.3f}"
                     f"  wait_inst/wave_cycles {m['SQ_WAIT_INST_ANY'] / max(m['SQ_WAVE_CYCLES'], 1):.3f}")
        else:
            line += f"  wait_any/wave_cycles {m['SQ_WAIT_ANY'] / max(m['SQ_WAVE_CYCLES'], 1):.3f}  TA busy (mean over 256 TAs / XCD cycles) {m['TA_TA_BUSY_sum'] / 256 / max(m['GRBM_GUI_ACTIVE'] / 8, 1):.3f}"
        print(line)
PY
