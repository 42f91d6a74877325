#!/bin/bash
# round 5, call O: a wave's first tickets wave-major (consecutive chunks to different workgroups) against workgroup-major: the switch point sweep and the 1 Mi sets; the renderer's deep-stack test; bench.py at the driver's flags
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05/spread.txt; rm -f $O
for lib in librodent_hip exp_spread; do
  echo "== $lib" >> $O
  RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 600 python scripts/threshold_sweep.py --scenes atrium 2>&1 | grep -v amdgpu.ids >> $O
  for rep in 1 2; do RODENT_HIP_LIB=$PWD/rodent_amd/lib/$lib.so timeout 300 python scripts/spill_experiment.py 2>&1 | grep -v amdgpu.ids >> $O; done
done
cat $O
timeout 900 python -m pytest tests -m gpu -x -q -k "deep_stacks_in_the_renderer" 2>&1 | tail -3
SECONDS=0
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err; echo "bench.py wall ${SECONDS} s"; tail -2 gpurun_out/r05/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_final.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step", "vs_baseline", "dtype")}, d["config"])
print({k: v for k, v in d["roofline"].items() if not isinstance(v, dict) and k not in ("what", "hbm_algorithmic_frac_is")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["extra"]["all_rays_bit_exact_vs_oracle"])
PY
