#!/bin/bash
# round 5, call AE: the whole -m gpu suite twice (flakes), smoke, bench.py at the driver's flags with its wall time
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
for i in 1 2; do SECONDS=0; timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_ae_$i.txt 2>&1; echo "suite run $i: $SECONDS s"; tail -3 gpurun_out/r05/tests_ae_$i.txt; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_ae.json 2> gpurun_out/r05/bench_ae.err; echo "bench wall $SECONDS s"; tail -2 gpurun_out/r05/bench_ae.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_ae.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step", "vs_baseline", "dtype")}, {k: v for k, v in d["config"].items() if "Mrays" in k or "Msamples" in k})
print({k: v for k, v in d["roofline"].items() if not isinstance(v, dict) and k not in ("what", "hbm_algorithmic_frac_is")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["extra"]["all_rays_bit_exact_vs_oracle"])
PY
