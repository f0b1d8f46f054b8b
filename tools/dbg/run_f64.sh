# float64 config-2 step: the one-launch response gradient's tests + kernel stats of the float64 step
cd /root/repo
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_spectral.py -q -m gpu -x -k "one_launch" 2>&1 | tail -15
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/stats64 -o r -- python $ROOT/bench.py --dtype f64 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/q/bench64.json 2> $ROOT/gpurun_out/q/bench64.err
cd $ROOT
rm -f gpurun_out/q/*/r_kernel_trace.csv
python - <<'PY'
import csv, json
for r in list(csv.DictReader(open("gpurun_out/q/stats64/r_kernel_stats.csv")))[:11]:
    print(f"  {r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.2f} us")
d = json.loads(open("gpurun_out/q/bench64.json").read().strip().splitlines()[-1])
print("f64", d["ms_per_step"], d["value"])
PY
