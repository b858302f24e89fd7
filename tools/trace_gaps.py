"""From a rocprofv3 --kernel-trace CSV: the period of a repeating sequence of kernels and the gaps between consecutive kernels on the device
(end of one to start of the next), so that what a call costs beyond the sum of its kernels' durations can be seen.
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [substring of the kernel that starts a call]"""
import csv
import glob
import statistics
import sys


def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "fx_frame_kernel"
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fxk::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    rows = rows[len(rows) // 3:]                       # the last passes: steady state
    gaps, durs = {}, {}
    for a, b in zip(rows, rows[1:]):
        gaps.setdefault((a[2][:44], b[2][:44]), []).append((b[0] - a[1]) / 1e3)
    for r in rows:
        durs.setdefault(r[2][:60], []).append((r[1] - r[0]) / 1e3)
    starts = [r[0] for r in rows if first in r[2]]
    periods = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
    print("period between launches of %s: median %.1f us, mean %.1f us over %d" % (first, statistics.median(periods), statistics.mean(periods), len(periods)))
    for k, v in sorted(durs.items()):
        print("  duration %-62s median %7.1f us  n %d" % (k, statistics.median(v), len(v)))
    for k, v in sorted(gaps.items()):
        print("  gap %-46s -> %-46s median %6.1f us  n %d" % (k[0], k[1], statistics.median(v), len(v)))


if __name__ == "__main__":
    main()
