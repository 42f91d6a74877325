#!/bin/bash
mkdir -p gpurun_out/r02 gpurun_out/profiles; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_refbuilt.py tests/test_gpu_parity.py tests/test_gpu_atrium.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --big --mid 2>&1 | tee gpurun_out/r02/sweep_top.log | cut -c1-200
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profiles/r02_top -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --only primary > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/profiles/r02_top -name "*kernel_stats.csv" | head -1 | xargs cut -c1-150 | head -8
