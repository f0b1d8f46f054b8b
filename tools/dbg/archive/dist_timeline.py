"""Kernel timeline of bench.py's replayed step with the gradient all-reduce in it (one rank over RCCL: BENCH_FORCE_DIST=1):
python tools/dbg/dist_timeline.py <kernel_trace.csv> [marker kernel]  -- step periods from the cascade-forward launches, one step's launches."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "sos_response_rc_fast"       # the first kernel of a step
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
per = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(idx, idx[1:])]
print("steps", len(idx), "periods us (every 10th):", [round(p, 1) for p in per[::10]])
# a step in the middle of the longest run of near-equal periods
k = len(idx) * 2 // 3
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev_end = max(prev_end, e)
