cd /root/repo
timeout 1800 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "kept_factors or fdn" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
