#!/bin/bash
# round 5, call F: the settled spill (test at the end of the step, error word in device memory, copied to the host page when the launch finishes) against round 4's library; the whole suite; host costs; bench.py
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
rm -f $O/spill_experiment2.txt
for rep in 1 2 3; do
  for lib in librodent_hip_r04 librodent_hip; do
    RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O/spill_experiment2.txt
  done
done
echo "== r04 library" > $O/host_call_costs.txt
RODENT_HIP_LIB=$PWD/rodent_amd/lib/librodent_hip_r04.so timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
echo "== r05 library" >> $O/host_call_costs.txt
timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -6 $O/tests_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_f.json 2> $O/bench_f.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_f.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step")}, d["extra"]["primary_kernel_ms"], d["extra"]["random_kernel_ms"], d["extra"]["random_Mrays_s"])
PY
cat $O/spill_experiment2.txt $O/host_call_costs.txt
