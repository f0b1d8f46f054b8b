# configs 3 / 4 quick loop: their tests, the replayed colorless training step (drop-in criteria against the torch lines) and the
# replayed FDN step, then the kernel table of the FDN bench (rocprofv3)
cd /root/repo
mkdir -p gpurun_out/q
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_round3_parity.py tests/test_hip_parity.py tests/test_objectives.py -q -m gpu -x -k "magnitude or sparsity or e7 or colorless or constant or objectives or mse or fdn16" 2>&1 | tail -4
for c in 0 1; do
FLAMO_TORCH_CRITERIA=$c python tools/train_colorless_fdn.py --steps 300 --graph --fused-adam 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('colorless, torch criteria $c:', round(d['ms_per_step'], 4), 'ms per step; losses', d['loss_last'])"
done
python tools/bench_fdn.py --dtype f32 2>/dev/null | tail -1 | cut -c1-300
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/c3 -o r -- python $ROOT/tools/bench_fdn.py --dtype f32 > /dev/null 2>&1
cd $ROOT
rm -f gpurun_out/q/*/r_kernel_trace.csv
python tools/dbg/kstats.py gpurun_out/q/c3/r_kernel_stats.csv 2>/dev/null | head -24
