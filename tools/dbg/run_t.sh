cd /root/repo
timeout 1800 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_round2_parity.py tests/test_round3_parity.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
