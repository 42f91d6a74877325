#!/bin/bash
# round 5, call U: after the pixel blocks -- render tests, the renderer's profiles again (they are stamped with the source hash), bench, the scene matrix with the tile statistic
mkdir -p gpurun_out/r05 gpurun_out/profiles; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -k "render or atrium or service or tiles" > gpurun_out/r05/tests_u.txt 2>&1; tail -3 gpurun_out/r05/tests_u.txt
TAG=r05
bash scripts/render_profile.sh $TAG cfg4 > gpurun_out/profiles/${TAG}_rp4.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp4.log
bash scripts/render_profile.sh $TAG cfg5 > gpurun_out/profiles/${TAG}_rp5.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp5.log
cp gpurun_out/profiles/${TAG}_render_profile_cfg4.json gpurun_out/profiles/${TAG}_render_profile_cfg5.json profiles/ 2>/dev/null
bash scripts/gpu_r05_scenes.sh > gpurun_out/r05/scenes.log 2>&1; cat gpurun_out/r05/scene_matrix.txt | cut -c1-400
SECONDS=0; timeout 900 python bench.py > gpurun_out/profiles/${TAG}_bench_line.json 2> gpurun_out/profiles/${TAG}_bench.err; echo "bench wall $SECONDS s"; tail -2 gpurun_out/profiles/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/profiles/${TAG}_bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["bound"], r["frac"], r["traffic"] and r["traffic"].get("bytes_per_launch"), d["extra"]["all_rays_bit_exact_vs_oracle"])
print(d["config"])
print("render", {k: {m: v[m].get("Msamples_s") for m in ("auto", "streaming", "megakernel") if isinstance(v.get(m), dict)} for k, v in d["extra"]["render"].items() if k.startswith("cfg")})
PY
