export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "deeper_than" 2>&1 | tail -15
