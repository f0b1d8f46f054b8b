cd /root/repo
timeout 1200 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "expm or matrix_exp or orth" 2>&1 | tail -3
bash tools/dbg/run_fdn.sh 2>&1 | tail -24
