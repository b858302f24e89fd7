"""Summarise a tools/profile.sh output directory: kernel-trace stats and per-kernel PMC sums."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def main():
    out = sys.argv[1]
    summary = {}
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        print("== kernel stats:", f)
        for row in csv.DictReader(open(f)):
            print("  %-60s calls %6s total_ns %14s avg_ns %12s pct %s" % (
                row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
            if "fx_frame_kernel" in row.get("Name", "") and "true, true" in row.get("Name", ""):
                summary["frame_kernel_avg_ns"] = float(row["AverageNs"])
                summary["frame_kernel_calls"] = int(row["Calls"])
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        durs = defaultdict(list)
        meta = {}
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            durs[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[n] = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        for n, d in durs.items():
            d2 = sorted(d)
            print("  trace %-50s n=%d min %.1f us median %.1f us max %.1f us  %s" % (n[:50], len(d), d2[0] / 1e3, d2[len(d2) // 2] / 1e3, d2[-1] / 1e3, meta[n]))
    pmc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in find(out, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            pmc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    for k in pmc:
        print("== PMC per launch (mean over dispatches):", k[:70])
        for c in sorted(pmc[k]):
            print("   %-28s %18.1f   (%d dispatches)" % (c, pmc[k][c] / cnt[k][c], cnt[k][c]))
        if "fx_frame_kernel" in k and "true, true" in k:
            summary["pmc_per_launch"] = {c: pmc[k][c] / cnt[k][c] for c in pmc[k]}
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
