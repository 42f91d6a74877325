#!/bin/bash
# round 6, call G: the new process test (miss records up front), smoke(), and the N = 2 bench on the one GPU (gloo, shared device)
export TMPDIR=/tmp; mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_render.py -m gpu -q -x -k "miss_records_up_front or through_indices" 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
RODENT_BENCH_BACKEND=gloo RODENT_BENCH_SHARE_GPUS=1 MASTER_PORT=29555 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --render-spp5 4 > gpurun_out/r06/bench_two_ranks_shared_gpu.json 2> gpurun_out/r06/bench_two_ranks.err; tail -2 gpurun_out/r06/bench_two_ranks.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_two_ranks_shared_gpu.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "scaling", "ms_per_step")}, d["config"]["workload"], d["config"]["predicted_scaling_x"]["strong_1Mi_primary_contiguous_ranges"], d["extra"]["strong_scaling_check"])
PY
