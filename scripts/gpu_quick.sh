export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cli" 2>&1 | tail -30
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.device_count())"
