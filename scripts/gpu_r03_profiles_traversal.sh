#!/bin/bash
# the traversal half of gpu_r03_profiles.sh (after a change to the traversal sources only), then bench.py
TAG=${1:-r03}; export TMPDIR=/tmp; mkdir -p gpurun_out/profiles
bash scripts/profile_round.sh $TAG > gpurun_out/profiles/${TAG}_round.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_round.log
bash scripts/profile_pmc.sh $TAG 0 > gpurun_out/profiles/${TAG}_pmc.log 2>&1; tail -2 gpurun_out/profiles/${TAG}_pmc.log
cp gpurun_out/profiles/${TAG}_traffic.json gpurun_out/profiles/${TAG}_pmc_counters.json profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/profiles/${TAG}_bench_line.json 2> gpurun_out/profiles/${TAG}_bench.err; tail -2 gpurun_out/profiles/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/profiles/${TAG}_bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["bound"], r["frac"], d["extra"]["with_schedule_history"]["primary_Mrays_s"], d["extra"]["all_rays_bit_exact_vs_oracle"], "valu_issue" in r["binding"])
PY
