#!/bin/bash
# SQ counters of the batch-walking row kernels (run on the GPU box): two passes of 8 counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_walk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq1 -o r -- python $ROOT/tools/dbg/walk_bench.py --reps 3 > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq2 -o r -- python $ROOT/tools/dbg/walk_bench.py --reps 3 > $OUT/sq2.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_walk"
for sub in ("sq1", "sq2"):
    f = glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for fn in f:
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
    for k, c in acc.items():
        if "walk" not in k and "spec_mid" not in k and "gradh" not in k: continue
        n = max(cnt[k], 1)
        print(f"{k:60s} n={n:3d} " + " ".join(f"{name[3:]}={v / n:.3g}" for name, v in sorted(c.items())))
PY
