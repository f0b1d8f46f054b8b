# matrix_exp tests + kernel durations inside the FDN step (rocprofv3), then the replayed steps without the profiler
cd /root/repo
mkdir -p gpurun_out/q
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "matrix_exp" 2>&1 | tail -3
ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/q/fdn -o r -- python $ROOT/tools/bench_fdn.py --dtype f32 > $ROOT/gpurun_out/q/fdn.json 2> $ROOT/gpurun_out/q/fdn.err
cd $ROOT
rm -f gpurun_out/q/*/r_kernel_trace.csv
grep -i "expm" gpurun_out/q/fdn/r_kernel_stats.csv | cut -d, -f1-4
python tools/bench_fdn.py --dtype f32 2>/dev/null | tail -1 | cut -c1-300
python tools/train_colorless_fdn.py --steps 300 --graph 2>/dev/null | tail -1 | cut -c1-260
