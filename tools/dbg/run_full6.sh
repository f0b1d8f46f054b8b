# all GPU tests (errors recorded) + the bench line (no CPU legs)
cd /root/repo
mkdir -p gpurun_out/full
FLAMO_RECORD_ERRORS=/root/repo/gpurun_out/full/achieved_errors.json timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee gpurun_out/full/tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
r = d["roofline"]
print({k: r[k] for k in ("frac", "launch_ms", "launches")}, r.get("events_in_step", {}).get("frac"), r.get("active_ms"))
for k, v in d.get("kernels", {}).items():
    print("  ", k, v.get("launch_ms"), v.get("frac_hbm_peak"))
for k in ("value_with_input_grad", "value_mse_objective", "value_generic_objective"):
    print(k, d.get(k))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("ms_per_step") if isinstance(v, dict) else v)
PY
