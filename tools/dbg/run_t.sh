cd /root/repo
timeout 2400 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_round2_parity.py tests/test_round3_parity.py tests/test_round4.py tests/test_objectives.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
bash tools/dbg/run_fdn.sh 2>&1 | tail -22
