#!/bin/bash
# the whole -m gpu suite + smoke (what the driver runs at round end), output kept short
export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
