#!/bin/bash
# end of round: what the driver runs -- the -m gpu suite, smoke(), bench.py at its flags
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/tests_final.txt 2>&1; tail -4 gpurun_out/r05/tests_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err; echo "bench wall $SECONDS s"; tail -2 gpurun_out/r05/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/bench_final.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step", "vs_baseline", "dtype")}, d["config"])
print({k: v for k, v in d["roofline"].items() if not isinstance(v, dict) and k not in ("what", "hbm_algorithmic_frac_is")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["extra"]["all_rays_bit_exact_vs_oracle"])
PY
