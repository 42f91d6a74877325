#!/bin/bash
# round 4 evidence: kernel traces + HBM traffic of bench.py (profile_round.sh), the twelve PMC groups on both ray sets
# (profile_pmc.sh), per-kernel profiles of the renderer on BASELINE configs 4 and 5 (render_profile.sh), then bench.py itself
TAG=${1:-r05}; export TMPDIR=/tmp; mkdir -p gpurun_out/profiles
bash scripts/profile_round.sh $TAG > gpurun_out/profiles/${TAG}_round.log 2>&1; tail -5 gpurun_out/profiles/${TAG}_round.log
bash scripts/profile_pmc.sh $TAG 0 > gpurun_out/profiles/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_pmc.log
# lane utilisation of the "refill" variant on the random set (VERDICT r2 item 3 asks for SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU before and after)
V=$(python -c "from rodent_amd import abi; print(abi.variants(2).index('refill'))")
timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/profiles/${TAG}_pmcrefill_random_sq1 -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-render --only random --variant $V > gpurun_out/profiles/${TAG}_pmcrefill_random_sq1.log 2>&1
python scripts/pmc_digest.py gpurun_out/profiles ${TAG}_pmcrefill k_bvh2 > gpurun_out/profiles/${TAG}_pmcrefill_digest.txt 2>&1; tail -9 gpurun_out/profiles/${TAG}_pmcrefill_digest.txt
bash scripts/render_profile.sh $TAG cfg4 > gpurun_out/profiles/${TAG}_rp4.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp4.log
bash scripts/render_profile.sh $TAG cfg5 > gpurun_out/profiles/${TAG}_rp5.log 2>&1; tail -3 gpurun_out/profiles/${TAG}_rp5.log
# the profiles must be in place (profiles/) for bench.py to quote them
cp gpurun_out/profiles/${TAG}_traffic.json gpurun_out/profiles/${TAG}_pmc_counters.json gpurun_out/profiles/${TAG}_render_profile_cfg4.json gpurun_out/profiles/${TAG}_render_profile_cfg5.json profiles/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/profiles/${TAG}_bench_line.json 2> gpurun_out/profiles/${TAG}_bench.err; tail -2 gpurun_out/profiles/${TAG}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/profiles/${TAG}_bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step")}, r["bound"], r["frac"], r["traffic"], r["l2_fabric_traffic"], d["extra"]["all_rays_bit_exact_vs_oracle"])
print("binding", {k: (v["frac"] if isinstance(v, dict) else v) for k, v in r["binding"].items()})
print("render", {k: {m: v[m].get("Msamples_s") for m in ("auto", "streaming", "megakernel") if isinstance(v.get(m), dict)} for k, v in d["extra"]["render"].items() if k.startswith("cfg")})
PY
ls gpurun_out/profiles | grep -v "^${TAG}_\(pmc_\|rp_\|fetch\|write\|tcc\|sq\|trace\|render\)" | head -30
