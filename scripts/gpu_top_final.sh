#!/bin/bash
# the default mapping "top": whole GPU suite, smoke, bench line, kernel traces + traffic, PMC groups, batch scaling, lab sweep of the top-image family
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_final.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu_final.log; tail -4 gpurun_out/r02/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench_stderr.log; tail -3 gpurun_out/r02/bench_stderr.log; python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['extra']['random_Mrays_s'], d['extra'].get('random_sorted',{}).get('Mrays_s'), d['extra'].get('with_schedule_history'), d['cpu_baseline']['value'], d['extra']['all_rays_bit_exact_vs_oracle'])"
bash scripts/profile_round.sh r02 > gpurun_out/r02/profile_round.log 2>&1; grep -A6 "only primary" gpurun_out/profiles/r02_digest.txt | cut -c1-150
bash scripts/profile_pmc.sh r02 0 > gpurun_out/r02/profile_pmc.log 2>&1; grep -c failed gpurun_out/r02/profile_pmc.log
python bench.py --steps 20 --warmup 5 2>/dev/null > gpurun_out/r02/bench_line_with_pmc.json; python -c "
import json; d=json.loads(open('gpurun_out/r02/bench_line_with_pmc.json').read().strip().splitlines()[-1]); print(d['value'], d['extra']['random_Mrays_s'], json.dumps(d['roofline']['binding'])[:1500]); print(json.dumps(d['roofline']['random']['binding'])[:1200])"
python scripts/batch_scaling.py 2>&1 | tee gpurun_out/r02/batch_scaling.txt | tail -12
RODENT_HIP_LAB=1 timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --big --only top15,top31w2,top63w4,top127w8,top255w16,top23,top15-keep,top127w8-keep,sorted-top63w4,top255p16-pf,top127p8,top63p4,top15p1,sorted-top255p16,top255p16-o16,top1023p16-o16,top255p8-o24,top63p4-o28,top255r16-32,top255r16-48,top-fused,top-prio64,top-prio96,top-prio128,fast,phased,sorted 2>&1 | tee gpurun_out/r02/sweep_top_family.log | cut -c1-130
RODENT_HIP_LAB=1 timeout 300 python scripts/lpt_experiment.py 2>&1 | tee gpurun_out/r02/lpt_experiment.txt | tail -9
RODENT_HIP_LAB=1 timeout 300 python scripts/trace_top.py 2>&1 | tee gpurun_out/r02/trace_top.txt | tail -3
