# all GPU tests as the driver runs them (-x), recording the achieved errors, then the profile collection
cd /root/repo
mkdir -p gpurun_out
FLAMO_RECORD_ERRORS=/root/repo/gpurun_out/achieved_errors.json timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/gputest.log; tail -8 gpurun_out/gputest.log
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; tail -3 gpurun_out/collect.log | cut -c1-300
