#!/bin/bash
# rocprofv3 kernel trace of the bench command; prints the per-kernel average over the graph replays
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
cd $ROOT
tail -1 $OUT/stats.log | cut -c1-200
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/trace_bench"
rows = list(csv.DictReader(open(glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed replays: take the last 40 % of the trace before the eager roofline leg is hard to delimit -> report medians
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); med = v2[len(v2) // 2]
    if len(v) >= 10:
        print(f"{len(v):5d} x  median {med:7.1f} us  mean {sum(v)/len(v):7.1f}  {k}")
PY
