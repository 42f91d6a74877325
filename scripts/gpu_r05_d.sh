#!/bin/bash
# round 5, call D (the reload test at the step start): spill with the rare paths out of line + pinned error flag + stateless default against round 4's library; the whole suite; bench.py
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
R04=$PWD/rodent_amd/lib/librodent_hip_r04.so
rm -f $O/ab_spill4.txt
for rep in 1 2 3; do
  echo "== r04 library, rep $rep" >> $O/ab_spill4.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/sweep_auto.py --steps 40 --variants top-nohint,top,refill 2>&1 | grep -v amdgpu.ids >> $O/ab_spill4.txt
  echo "== r05 library, rep $rep" >> $O/ab_spill4.txt
  timeout 600 python scripts/sweep_auto.py --steps 40 --variants top-nohint,top,refill 2>&1 | grep -v amdgpu.ids >> $O/ab_spill4.txt
done
for rep in 1 2; do
  echo "== r04 library, rep $rep" >> $O/ab_spill4.txt
  RODENT_HIP_LIB=$R04 timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/ab_spill4.txt
  echo "== r05 library, rep $rep" >> $O/ab_spill4.txt
  timeout 600 python scripts/frame_rate.py --spp 64 2>&1 | tail -1 >> $O/ab_spill4.txt
done
echo "== r04 library" > $O/host_call_costs.txt
RODENT_HIP_LIB=$R04 timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
echo "== r05 library" >> $O/host_call_costs.txt
timeout 300 python scripts/host_call_costs.py 2>&1 | grep -v amdgpu.ids >> $O/host_call_costs.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_all.txt 2>&1
tail -6 $O/tests_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_d.json 2> $O/bench_d.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_d.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step")}, d["config"], d["extra"]["primary_kernel_ms"], d["extra"]["random_kernel_ms"], d["extra"].get("random_with_kind_hint"))
PY
cat $O/ab_spill4.txt $O/host_call_costs.txt
