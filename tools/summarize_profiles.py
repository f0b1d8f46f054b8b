#!/usr/bin/env python
"""gpurun_out/final/ (written by tools/collect_profiles.sh on the GPU box) -> profiles/<tag>_*:
the bench line, the rocprofv3 --stats kernel table of the same command, per-kernel HBM traffic from
the two PMC passes (FETCH_SIZE doubled on gfx950, see MI355X_MICROARCH.md), and the FDN numbers."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")


def pmc(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
            acc[(r["Kernel_Name"], grid)].append(float(r["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01z"
    os.makedirs(DST, exist_ok=True)
    bench = json.load(open(os.path.join(SRC, "bench.json")))
    json.dump(bench, open(os.path.join(DST, f"{tag}_bench.json"), "w"), indent=1)
    shutil.copy(os.path.join(SRC, "stats", "r_kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_kernel_stats_rocprofv3.csv"))
    if os.path.exists(os.path.join(SRC, "fdn_stats", "r_kernel_stats.csv")):
        shutil.copy(os.path.join(SRC, "fdn_stats", "r_kernel_stats.csv"), os.path.join(DST, f"{tag}_fdn_kernel_stats_rocprofv3.csv"))
    for sub, name in (("c5_stats", "config5"), ("c4_stats", "colorless")):
        if os.path.exists(os.path.join(SRC, sub, "r_kernel_stats.csv")):
            shutil.copy(os.path.join(SRC, sub, "r_kernel_stats.csv"), os.path.join(DST, f"{tag}_{name}_kernel_stats_rocprofv3.csv"))
    mf = os.path.join(SRC, "c5_pmc", "r_counter_collection.csv")
    if os.path.exists(mf):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(mf)):
            if "fl::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(DST, f"{tag}_config5_pmc_sq.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
            wr.writerow(["kernel", "launches"] + [n + "_per_launch" for n in names])
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
                n = max(len(x) for x in v.values())
                wr.writerow([k[:120], n] + ["%.0f" % (sum(v.get(c, [0])) / max(len(v.get(c, [1])), 1)) for c in names])
    fdn = {}
    for name in ("fdn_b1", "fdn_b8", "config5", "colorless", "colorless_graph"):
        p = os.path.join(SRC, name + ".json")
        if os.path.exists(p) and os.path.getsize(p):
            fdn[name] = json.load(open(p))
    json.dump(fdn, open(os.path.join(DST, f"{tag}_fdn_bench.json"), "w"), indent=1)
    for name in ("hbm_probe.txt", "hbm_probe.json", "policy_ab.txt"):
        if os.path.exists(os.path.join(SRC, name)) and os.path.getsize(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{tag}_{name}"))
    fetch = pmc(os.path.join(SRC, "pmc_fetch", "r_counter_collection.csv"), "FETCH_SIZE")
    write = pmc(os.path.join(SRC, "pmc_write", "r_counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if "fl::" not in name:
            continue
        f = fetch.get(key, [])
        w = write.get(key, [])
        fb = 2 * 1024 * sum(f) / len(f) if f else 0.0       # KB -> bytes, x2 on gfx950
        wb = 1024 * sum(w) / len(w) if w else 0.0
        rows.append((name, grid, len(f), fb, wb, fb + wb))
    with open(os.path.join(DST, f"{tag}_pmc_hbm_traffic.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "grid_size", "launches", "fetch_bytes_per_launch(2*FETCH_SIZE*1024)", "write_bytes_per_launch(WRITE_SIZE*1024)", "total_bytes_per_launch"])
        for r in rows:
            wr.writerow([r[0], r[1], r[2], f"{r[3]:.0f}", f"{r[4]:.0f}", f"{r[5]:.0f}"])
    # what bench.py reads back as roofline.traffic: bytes per launch of the dominant kernels, keyed by bench.py's kernel tags
    def largest(pat):
        c = [r for r in rows if pat in r[0]]
        c.sort(key=lambda r: -r[5])
        return c[0] if c else None
    tags = {"spec_mid_walk[8->8,spec]": "spec_mid_walk<16, 15, 8, 8", "spec_gradh_walk": "spec_gradh_walk<16, 15, 8, 8",
            "spec_mid[8->8,H,inv,spec]": "spec_mid<16, 15, 8, 8, true, true", "spec_mid[8->8,spec]": "spec_mid<16, 15, 8, 8, false, false",
            "spec_cols_fwd": "spec_cols_fwd<", "spec_cols_fwd+response": "cols_fwd_rc_kernel<", "spec_cols_inv": "spec_cols_inv<", "spec_cols_inv+grad_cols": "spec_cols_inv<8, 25, 16, 1, true, true>", "mimo_gradh[cols=32,8x8]": "mimo_gradh_kernel<float, 4, 4>",
            "sos_response_rc": "sos_response_rc_ba_kernel", "sos_response_bwd_rc": "sos_bwd_lanes_kernel<float, 8, 8", "mimo_full": "mimo_full_kernel<float, 8, 4, 1, false>"}
    traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3`, FETCH_SIZE doubled per "
                         f"MI355X_MICROARCH.md; profiles/{tag}_pmc_hbm_traffic.csv"}
    try:        # which code the counters were taken on: the hash and whether the tree was clean when this summary was written
        import subprocess
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "flamo_amd", "bench.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
        traffic["commit"] = head + ("+uncommitted changes" if dirty else "")
    except Exception:           # noqa: BLE001
        traffic["commit"] = None
    # average launch duration of the same kernels in the rocprofv3 --kernel-trace --stats run of the bench command (graph replays)
    stats = {}
    try:
        for r in csv.DictReader(open(os.path.join(SRC, "stats", "r_kernel_stats.csv"))):
            stats[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    except OSError:
        pass
    for k, pat in tags.items():
        r = largest(pat)
        if r:
            traffic[k] = {"bytes_per_launch": r[5], "fetch_bytes": r[3], "write_bytes": r[4], "launches": r[2], "kernel": r[0][:100]}
            st = [v for n, v in stats.items() if n[:100] == r[0][:100]]
            if st:
                traffic[k]["rocprofv3_avg_launch_us"] = st[0][0] / 1e3
                traffic[k]["rocprofv3_calls"] = st[0][1]
    json.dump(traffic, open(os.path.join(DST, "pmc_hbm_traffic.json"), "w"), indent=1)
    sq = os.path.join(SRC, "bench_sq", "r_counter_collection.csv")
    if os.path.exists(sq):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(sq)):
            if "fl::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
        with open(os.path.join(DST, f"{tag}_bench_pmc_sq.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["kernel", "launches"] + [n + "_per_launch" for n in names] + ["valu_per_wave_cycle", "lds_conflict_per_active"])
            for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
                n = max(len(x) for x in v.values())
                m = {c: (sum(v.get(c, [0])) / max(len(v.get(c, [1])), 1)) for c in names}
                wr.writerow([k[:120], n] + ["%.0f" % m[c] for c in names] +
                            ["%.3f" % (m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_WAVE_CYCLES"], 1)), "%.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1))])
    big = [r for r in rows if "spec_mid" in r[0]]
    big.sort(key=lambda r: -r[5])
    if big:
        print("dominant kernel traffic (largest spec_mid launch):", big[0][0][:60], "grid", big[0][1], "bytes %.4g" % big[0][5])
    print("value %.4g %s, %.4f ms/step; roofline frac %.3f" % (bench["value"], bench["unit"], bench["ms_per_step"],
                                                               (bench.get("roofline") or {}).get("frac", float("nan"))))


if __name__ == "__main__":
    main()
