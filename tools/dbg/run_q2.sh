cd /root/repo
mkdir -p gpurun_out/q2
for rep in 1 2 3; do for m in 0 0x1030; do
  FLAMO_STREAM_POLICY=$m timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('policy=$m', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 4), 'slot', round(r['launch_ms'], 5), 'active', round(r.get('active_ms', 0), 5), 'events', round(r.get('events_in_step', {}).get('launch_ms', 0), 5))"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/q2/stats -o r -- python /root/repo/bench.py --no-cpu-baseline --no-extras > /root/repo/gpurun_out/q2/bench.json 2> /root/repo/gpurun_out/q2/bench.err
cd /root/repo; rm -f gpurun_out/q2/stats/r_kernel_trace.csv
python tools/dbg/kstats.py gpurun_out/q2/stats/r_kernel_stats.csv | head -9
python -c "
import json
d = json.loads(open('gpurun_out/q2/bench.json').read().strip().splitlines()[-1]); r = d['roofline']; print('under rocprof:', d['ms_per_step'], r['frac'], r['launch_ms'], r.get('active_ms'))"
