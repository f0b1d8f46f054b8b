# launch pair: timeline of the roles, its tests, the replayed step
cd /root/repo
timeout 300 python tools/dbg/pair_timeline.py --reps 2 2>&1 | grep -v "t= " | tail -14
timeout 900 python -m pytest tests/test_spectral.py tests/test_cascade2.py -q -m gpu -x -k "launch_pair or cascade or rc" 2>&1 | tail -3
for p in 1 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
