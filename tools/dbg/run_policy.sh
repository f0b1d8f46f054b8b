# cache-policy sweep of the pipeline's streams on the replayed config-2 step + the spectral / cascade tests under the default
cd /root/repo
mkdir -p gpurun_out/policy
timeout 900 python -m pytest tests/test_spectral.py tests/test_cascade2.py -q -m gpu -x 2>&1 | tail -3
timeout 1500 python tools/dbg/policy_sweep.py --greedy 2>&1 | grep -v Warning | tee gpurun_out/policy/sweep.txt
