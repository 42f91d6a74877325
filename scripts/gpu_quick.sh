export TMPDIR=/tmp; mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py tests/test_refbuilt.py -m gpu -x -q 2>&1 | tail -6
timeout 900 python scripts/sweep_widths.py --widths 2,4,8 --all-variants --big --only fast,single,top 2>&1 | tee gpurun_out/r03/sweep_widths.log | cut -c1-200
