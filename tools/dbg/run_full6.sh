# all GPU tests + the bench line (no CPU legs)
cd /root/repo
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/full/tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
r = d["roofline"]
print({k: r[k] for k in ("frac", "launch_ms", "launches")}, r.get("events_in_step", {}).get("frac"), r.get("measured"))
print(d["device"])
for k, v in d.get("secondary", {}).items():
    print(k, v.get("ms_per_step") if isinstance(v, dict) else v)
PY
