#!/bin/bash
# round 6 evidence on the round's final sources: the whole GPU suite, then kernel traces + HBM traffic of bench.py, the PMC groups on both ray sets,
# per-kernel profiles of the renderer on BASELINE configs 4 and 5, and bench.py itself (gpu_r05_profiles.sh with the round's tag)
export TMPDIR=/tmp; mkdir -p gpurun_out/profiles
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6
bash scripts/gpu_r05_profiles.sh r06
