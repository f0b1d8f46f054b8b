"""Print the last full step of a rocprofv3 kernel trace csv: python tools/dbg/step_timeline.py <csv> <marker substring>
(steps are delimited by every second-to-last..last occurrence of the marker kernel; with two markers per step pass 2)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0, i1 = idx[-1 - 2 * per], idx[-1 - per]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  {r['Kernel_Name'][:110]}")
print(i1 - i0, "launches", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
