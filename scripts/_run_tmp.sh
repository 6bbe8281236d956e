timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -q -m gpu 2>&1 | tail -40
timeout 300 python scripts/time_backward.py --config 2 2>&1 | tail -3
