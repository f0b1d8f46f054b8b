# quick loop: spectral tests, the bench line without the CPU legs, kernel stats of the f32 and f64 steps
cd /root/repo
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_spectral.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/q/test.log; tail -3 gpurun_out/q/test.log
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $ROOT/gpurun_out/q/bench.json 2> $ROOT/gpurun_out/q/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/stats64 -o r -- python $ROOT/bench.py --dtype f64 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/q/bench64.json 2> $ROOT/gpurun_out/q/bench64.err
cd $ROOT
rm -f gpurun_out/q/*/r_kernel_trace.csv
python - <<'PY'
import csv, json
for tag in ("stats", "stats64"):
    print(tag)
    for r in list(csv.DictReader(open(f"gpurun_out/q/{tag}/r_kernel_stats.csv")))[:14]:
        print(f"  {r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.2f} us")
for f in ("bench", "bench64"):
    try:
        d = json.loads(open(f"gpurun_out/q/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "no line", e)
PY
