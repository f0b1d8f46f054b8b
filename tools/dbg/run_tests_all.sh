# all GPU tests WITHOUT -x (every failure in one pass), recording the achieved errors
cd /root/repo
mkdir -p gpurun_out
FLAMO_RECORD_ERRORS=/root/repo/gpurun_out/achieved_errors.json timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/gputest_all.log; tail -12 gpurun_out/gputest_all.log
