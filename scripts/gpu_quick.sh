export TMPDIR=/tmp; mkdir -p gpurun_out/r03
RODENT_HIP_LAB=1 timeout 900 python scripts/sweep_widths.py --widths 2 --all-variants --steps 30 --only top-two,top-partner-48-24,top-partner-32-40,top-partner-64-16,top-partner-999-64 2>&1 | tee gpurun_out/r03/sweep_partner_order.log | cut -c1-160
RODENT_HIP_LAB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or bijection or deep_stack" 2>&1 | tail -3
