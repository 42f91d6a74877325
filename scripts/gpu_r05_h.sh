#!/bin/bash
# round 5, call H: joint fetches everywhere (BVH2 persistent kernels, the renderer's traversal kernels): suite, A/B against round 4's library, frames, scene matrix, bench
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -6 $O/tests_all.txt
rm -f $O/joint_loads2.txt
R04=$PWD/rodent_amd/lib/librodent_hip_r04.so
for rep in 1 2 3; do
  for lib in librodent_hip_r04 librodent_hip; do
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O/joint_loads2.txt
  done
done
for rep in 1 2; do
  echo "== r04 library, rep $rep" >> $O/joint_loads2.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/joint_loads2.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 2>&1 | tail -1 >> $O/joint_loads2.txt
  echo "== r05 library, rep $rep" >> $O/joint_loads2.txt
  timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/joint_loads2.txt
  timeout 600 python scripts/frame_rate.py --scene cornell --size 1920x1080 --spp 64 --len 4 2>&1 | tail -1 >> $O/joint_loads2.txt
done
cat $O/joint_loads2.txt
timeout 2400 python scripts/scene_matrix.py --json $O/scene_matrix.json 2>&1 | grep -v amdgpu.ids > $O/scene_matrix.txt
cat $O/scene_matrix.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_h.json 2> $O/bench_h.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_h.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step")}, d["config"])
print({k: v for k, v in d["roofline"].items() if not isinstance(v, dict)})
PY
