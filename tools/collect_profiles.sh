#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench line with the CPU baseline, the
# rocprofv3 kernel stats of the same bench command, the two PMC passes for HBM traffic, and the
# FDN (config 3) numbers.  Everything lands in gpurun_out/final/; tools/summarize_profiles.py
# turns it into the files committed under profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fdn_stats -o r -- python $ROOT/tools/bench_fdn.py --dtype f32 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o r -- python $ROOT/tools/bench_fdn.py --workload config5 --dtype f32 --steps 5 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o r -- python $ROOT/tools/train_colorless_fdn.py --steps 20 --warmup 2 > /dev/null 2>&1
# SQ activity of the bench step's kernels (its own pass: counters only)
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/bench_sq -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
# matrix-core activity of the config-5 chain (its own pass: counters only)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/c5_pmc -o r -- python $ROOT/tools/bench_fdn.py --workload config5 --dtype f32 --steps 2 > /dev/null 2>&1
cd $ROOT
python tools/bench_fdn.py --workload config5 --dtype f32 --steps 8 2>/dev/null | tail -1 > $OUT/config5.json
python tools/train_colorless_fdn.py --steps 50 2>/dev/null | tail -1 > $OUT/colorless.json
python tools/train_colorless_fdn.py --steps 200 --graph 2>/dev/null | tail -1 > $OUT/colorless_graph.json
python tools/bench_fdn.py 2>/dev/null | tail -1 > $OUT/fdn_b1.json
python tools/bench_fdn.py --batch 8 2>/dev/null | tail -1 > $OUT/fdn_b8.json
rm -f $OUT/*/r_kernel_trace.csv.bak
ls -la $OUT $OUT/stats | head -30
cut -c1-400 $OUT/bench.json
# round 6: the memory system's ceilings on the library's own streaming kernels, and the cache-policy A/B of the step
python tools/dbg/hbm_probe.py --json $OUT/hbm_probe.json > $OUT/hbm_probe.txt 2>&1
python tools/dbg/policy_ab.py 0x0,0x30,0x34,0x1030,0x100 3 2>/dev/null | grep mask > $OUT/policy_ab.txt
