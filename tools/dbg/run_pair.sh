# launch pair (fusedfwd.hip): its test, the spectral / objective tests, the replayed step with the pair on and off
cd /root/repo
mkdir -p gpurun_out/pair
timeout 1200 python -m pytest tests/test_spectral.py tests/test_objectives.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -4
for p in 1 0 1 0; do
  FLAMO_LAUNCH_PAIR=$p timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$p', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pair/stats -o r -- python /root/repo/bench.py --no-cpu-baseline --no-extras > /root/repo/gpurun_out/pair/bench.json 2> /root/repo/gpurun_out/pair/bench.err
cd /root/repo; rm -f gpurun_out/pair/stats/r_kernel_trace.csv
python tools/dbg/kstats.py gpurun_out/pair/stats/r_kernel_stats.csv | head -16
