#!/bin/bash
# SQ counters of the fused-pipeline kernels (run on the GPU box): LDS conflicts, VALU activity, wait states
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_spec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o r -- python $ROOT/tools/dbg/spec_bench.py --steps 2 > $OUT/sq.log 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_spec"
f = glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"][:70]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        if row["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in acc.items():
    if "spec_" not in k and "mimo" not in k and "sos" not in k: continue
    n = max(cnt[k], 1)
    wc = c["SQ_WAVE_CYCLES"]
    print(f"{k:70s} n={n:3d} lds_conf/active={c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.2f} "
          f"valu/wave={c['SQ_ACTIVE_INST_VALU']/max(wc,1):.2f} any/wave={c['SQ_ACTIVE_INST_ANY']/max(wc,1):.2f} wait_any/wave={c['SQ_WAIT_ANY']/max(wc,1):.2f} "
          f"wait_inst/wave={c['SQ_WAIT_INST_ANY']/max(wc,1):.2f} lds_active/busy={c['SQ_LDS_IDX_ACTIVE']/max(c['SQ_BUSY_CYCLES'],1):.2f}")
PY
