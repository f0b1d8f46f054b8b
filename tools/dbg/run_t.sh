cd /root/repo
timeout 1800 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "kept_factors or fdn" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
bash tools/dbg/run_fdn.sh 2>&1 | grep -E "passed|failed|colorless|workload|solve|dud|mimo_full" | head -12
