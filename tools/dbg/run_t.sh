cd /root/repo
timeout 900 python -m pytest tests/test_cascade2.py tests/test_spectral.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 4), d['roofline']['frac'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/q2/stats -o r -- python /root/repo/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
cd /root/repo; python tools/dbg/kstats.py gpurun_out/q2/stats/r_kernel_stats.csv | head -9
