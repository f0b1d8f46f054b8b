cd /root/repo
mkdir -p gpurun_out
FLAMO_RECORD_ERRORS=/root/repo/gpurun_out/achieved_errors.json timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/gputest_full.log
grep -E "passed|failed|error" gpurun_out/gputest_full.log | tail -5
grep -E "^(FAILED|ERROR)|assert|Error" gpurun_out/gputest_full.log | head -40
