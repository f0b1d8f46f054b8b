#!/bin/bash
# rocprofv3 kernel trace of bench.py's replayed step: per kernel of the step, its mean duration and the mean gap to its successor
# over the replays (consecutive occurrences of the step's launch sequence).   run on the GPU box: bash tools/dbg/step_gaps.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/step_gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o r -- python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 40 > $OUT/log.txt 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/step_gaps"
rows = list(csv.DictReader(open(glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first = [i for i, n in enumerate(names) if "cols_fwd_rc_kernel" in n]
# a replayed step: from one pair launch to the next with exactly 8 launches in between
import collections as _c
per = _c.Counter(b - a for a, b in zip(first, first[1:])).most_common(1)[0][0]
steps = [(a, b) for a, b in zip(first, first[1:]) if b - a == per]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); tot = []
for a, b in steps:
    tot.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
    for i in range(a, b):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        dur[i - a].append(e - s)
        gap[i - a].append(int(rows[i + 1]["Start_Timestamp"]) - e)
tot.sort()
print(len(steps), "replayed steps; step median %.1f us, min %.1f" % (tot[len(tot) // 2] / 1e3, tot[0] / 1e3))
a = steps[len(steps) // 2][0]
for k in range(per):
    d, g = sorted(dur[k]), sorted(gap[k])
    print("%-60s dur median %6.1f min %6.1f | gap to next median %5.2f min %5.2f" % (names[a + k][:60], d[len(d) // 2] / 1e3, d[0] / 1e3, g[len(g) // 2] / 1e3, g[0] / 1e3))
PY
