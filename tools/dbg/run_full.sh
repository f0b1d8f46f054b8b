cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/gputest.log; tail -6 gpurun_out/gputest.log
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -5 gpurun_out/collect.log
