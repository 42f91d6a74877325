export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 900 python scripts/joint_sweep.py 2>&1 | tee gpurun_out/r03/joint_sweep.txt
