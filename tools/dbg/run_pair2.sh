cd /root/repo
timeout 600 python -m pytest tests/test_spectral.py -q -m gpu -x -k "launch_pair" 2>&1 | tail -3
for rep in 1 2; do
for d in 100 200 400 1000 100000 50; do
  FLAMO_PAIR_DENSITY=$d timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('density=$d', d['ms_per_step'], d['value'])"
done
done
