#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command line (run on the GPU box); prints one step's timeline between two
# occurrences of a marker kernel.   usage: trace_cmd.sh <marker substring> <python args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_cmd
rm -rf $OUT; mkdir -p $OUT
MARK="$1"; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o r -- python "$@" > $OUT/log.txt 2>&1
cd $ROOT
MARK="$MARK" python - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/trace_cmd"
rows = list(csv.DictReader(open(glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = os.environ["MARK"]
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# consecutive marker occurrences that are far apart delimit steps: take the last full step
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 5]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:6.1f}  {r['Kernel_Name'][:120]}")
print(i1 - i0, "launches,", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
PY
