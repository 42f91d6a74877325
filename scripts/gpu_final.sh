#!/bin/bash
# end-of-round evidence: the whole GPU suite, smoke(), the bench line, the round's kernel traces + traffic passes, renderer rates
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_final.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu_final.log; tail -4 gpurun_out/r02/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench_stderr.log; tail -3 gpurun_out/r02/bench_stderr.log; python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['extra']['random_Mrays_s'], d['extra'].get('random_sorted',{}).get('Mrays_s'), d['extra'].get('with_schedule_history',{}).get('primary_Mrays_s'), d['cpu_baseline']['value'], d['extra']['all_rays_bit_exact_vs_oracle']); print(json.dumps(d['roofline']['binding'])[:1800])"
bash scripts/profile_round.sh r02 > gpurun_out/r02/profile_round.log 2>&1; grep -A6 "only primary" gpurun_out/profiles/r02_digest.txt | cut -c1-150
bash scripts/gpu_render.sh 2>&1 | tail -34
