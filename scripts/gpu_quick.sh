#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/profiles
timeout 900 python bench.py > gpurun_out/profiles/r03_bench_line.json 2> gpurun_out/profiles/r03_bench.err; tail -2 gpurun_out/profiles/r03_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles/r03_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["extra"]["random_8Mi_rays_per_launch"], d["extra"]["primary_16Mi_rays_per_launch"])
print({k: v["auto"]["Msamples_s"] for k, v in d["extra"]["render"].items() if k.startswith("cfg")})
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k bench 2>&1 | tail -2
