#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them at round end
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05/tests_all.txt 2>&1; tail -16 gpurun_out/r05/tests_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
