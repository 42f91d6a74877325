#!/bin/bash
# round 6, call K (last): the GPU suite on the final tree with the product library, the Cornell / atrium parity tests with the lab library (every
# order-preserving row of its variant tables), smoke(), bench.py as the driver runs it
export TMPDIR=/tmp; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
RODENT_HIP_LAB=1 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py -m gpu -q -x -k "golden or atrium_sample or all_benchmark_rays or ragged" 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "scaling")}, r["bound"], r["frac"], r["hbm"], d["extra"]["all_rays_bit_exact_vs_oracle"], d["cpu_baseline"]["value"], d["config"]["cfg5_Msamples_s"], d["config"]["cfg4_Msamples_s"])
PY
