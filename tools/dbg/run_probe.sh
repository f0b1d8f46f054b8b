# HBM ceiling probe (fl_hbm_probe) + the quick loop's bench line
cd /root/repo
mkdir -p gpurun_out/probe
timeout 600 python tools/dbg/hbm_probe.py --json gpurun_out/probe/hbm_probe.json 2>&1 | tee gpurun_out/probe/hbm_probe.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/probe/bench.json 2> gpurun_out/probe/bench.err; tail -c 600 gpurun_out/probe/bench.json
