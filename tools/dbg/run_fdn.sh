cd /root/repo
mkdir -p gpurun_out/q
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels.py tests/test_round3_parity.py -q -m gpu -x -k "transform or fft or rfft or fdn or any_length or colorless" 2>&1 | tail -4
python tools/train_colorless_fdn.py --steps 300 --graph --fused-adam 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('colorless:', round(d['ms_per_step'], 4), 'ms per step; losses', d['loss_last'])"
python tools/bench_fdn.py --dtype f32 2>/dev/null | tail -1 | cut -c1-300
python tools/bench_fdn.py --dtype f32 --batch 8 2>/dev/null | tail -1 | cut -c1-300
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/c3 -o r -- python $ROOT/tools/bench_fdn.py --dtype f32 > /dev/null 2>&1
cd $ROOT
rm -f gpurun_out/q/*/r_kernel_trace.csv
python - <<'PY'
import csv
for r in csv.DictReader(open("/root/repo/gpurun_out/q/c3/r_kernel_stats.csv")):
    if 'fft_' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
