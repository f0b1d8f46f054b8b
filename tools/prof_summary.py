"""Summarise a rocprofv3 results database (rocpd sqlite) into a small CSV:
    python tools/prof_summary.py gpurun_out/prof1/r1_results.db profiles/r01_bench_kernel_stats.csv
Columns: kernel, calls, total_us, avg_us, pct  (same numbers as `rocprofv3 --stats`)."""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)            # drop the argument list
    name = name.replace("void ", "")
    return name[:140]


def main(db, out):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, total, avg, pct in rows:
            w.writerow([short(name), calls, f"{total / 1e3:.1f}" if total > 1e6 else f"{total:.1f}", f"{avg / 1e3:.2f}" if total > 1e6 else f"{avg:.2f}", f"{pct:.2f}"])
    print(f"wrote {out} ({len(rows)} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
