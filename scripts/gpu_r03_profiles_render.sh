#!/bin/bash
# the renderer half of gpu_r03_profiles.sh (after a change to the renderer sources only), then bench.py
TAG=${1:-r03}; export TMPDIR=/tmp; mkdir -p gpurun_out/profiles
bash scripts/render_profile.sh $TAG cfg4 > gpurun_out/profiles/${TAG}_rp4.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp4.log
bash scripts/render_profile.sh $TAG cfg5 > gpurun_out/profiles/${TAG}_rp5.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp5.log
cp gpurun_out/profiles/${TAG}_render_profile_cfg4.json gpurun_out/profiles/${TAG}_render_profile_cfg5.json profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/profiles/${TAG}_bench_line.json 2> gpurun_out/profiles/${TAG}_bench.err; tail -2 gpurun_out/profiles/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/profiles/${TAG}_bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["bound"], r["frac"], d["extra"]["with_schedule_history"]["primary_Mrays_s"], d["extra"]["all_rays_bit_exact_vs_oracle"], "valu_issue" in r["binding"])
for k, v in d["extra"]["render"].items():
    if k.startswith("cfg"): print(k, {m: v[m].get("Msamples_s") for m in ("auto", "streaming", "streaming_sorted", "megakernel") if isinstance(v.get(m), dict)}, v["auto_mapping"], "profile ok" if "not_quoted" not in v["per_kernel_profiled"] else v["per_kernel_profiled"])
PY
