export TMPDIR=/tmp; mkdir -p gpurun_out/r03
export RODENT_HIP_LAB=1
for v in top top-one top-lazy-one; do
  idx=$(python -c "from rodent_amd import abi; print(abi.variants(2).index('$v'))")
  for rep in 1 2; do python bench.py --variant $idx --no-cpu-baseline --steps 100 --warmup 10 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['extra']['primary_kernel_ms']['mean'], d['extra']['random_Mrays_s'])"; done
done
