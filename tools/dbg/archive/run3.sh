cd /root/repo
ROOT=/root/repo
OUT=$ROOT/gpurun_out/r05a
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fdn_stats -o r -- python $ROOT/tools/bench_fdn.py --dtype f32 > $OUT/fdn.log 2>&1
cd $ROOT
head -30 $OUT/stats/*/r_kernel_stats.csv | cut -c1-220
echo ----
head -30 $OUT/fdn_stats/*/r_kernel_stats.csv | cut -c1-220
rm -f $OUT/*/*/r_kernel_trace.csv $OUT/*/*/*.db
