export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 600 python scripts/sort_sweep.py 2>&1 | tee gpurun_out/r03/sort_sweep.txt
