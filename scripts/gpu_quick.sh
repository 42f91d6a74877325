export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 900 python scripts/bounce_coherence_experiment.py 2>&1 | tee gpurun_out/r03/bounce_coherence_experiment.txt | tail -40
