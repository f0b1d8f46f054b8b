cd /root/repo
for rep in 1 2 3; do
for d in 100 120 140 170; do
  FLAMO_PAIR_DENSITY=$d timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('density=$d', round(d['ms_per_step']*1e3,1), round(d['ms_per_step_steady']['ms_per_step']*1e3,1))"
done
done
