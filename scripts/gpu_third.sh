#!/bin/bash
mkdir -p gpurun_out/r02; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu3.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu3.log; tail -6 gpurun_out/r02/pytest_gpu3.log
timeout 600 python bench.py > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench_stderr.log; cat gpurun_out/r02/bench_line.json; tail -3 gpurun_out/r02/bench_stderr.log
timeout 300 python bench.py --variant 2 --no-cpu-baseline > gpurun_out/r02/bench_line_phased.json 2>&1; cat gpurun_out/r02/bench_line_phased.json
bash scripts/profile_round.sh r02 > gpurun_out/r02/profile_round.log 2>&1; tail -40 gpurun_out/r02/profile_round.log
