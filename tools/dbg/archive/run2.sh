cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r05a_gputest.log; tail -8 gpurun_out/r05a_gputest.log
timeout 600 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r05a_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k, v in d['kernels'].items(): print(k, v.get('launch_ms'))
print({k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in d['secondary'].items()})
PY
