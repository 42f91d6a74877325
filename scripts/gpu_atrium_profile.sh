mkdir -p gpurun_out/profiles gpurun_out/r02; export TMPDIR=/tmp
python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')"
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profiles/r02_atrium -o rodent -- rodent_amd/bin/rodent $A > gpurun_out/profiles/r02_atrium.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/profiles/r02_atrium/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f} %")
PY
RODENT_HIP_LAB=1 timeout 300 python scripts/sweep.py --width 2 --variants 0,20 2>&1 | tail -3
