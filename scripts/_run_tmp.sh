timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|max err|FAILED" | head -30
