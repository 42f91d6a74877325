export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 600 python scripts/split_experiment.py 2>&1 | tee gpurun_out/r03/split_experiment.txt
