"""Timeline view of a rocprofv3 --kernel-trace CSV (MEASUREMENT TOOL): where the device time of ONE repetition goes.

    rocprofv3 --kernel-trace -f csv -d DIR -- python tools/prof_target.py bf16 32 4
    python tools/trace_timeline.py DIR MARKER [--list]

A repetition starts at every dispatch whose kernel name contains MARKER (e.g. prep_nhwc4_bf16, the first launch of a bf16
forward or training step); for the last one it prints the wall span from first start to last end, the union of busy intervals, the idle
time between dispatches, the per-kernel totals, and (--list) every dispatch with the gap in front of it."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    p = n.find("(")
    return (n[:p] if p > 0 else n)[:78]


def main():
    d, marker = sys.argv[1], sys.argv[2]
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + d)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 2:
        raise SystemExit("marker %r matches %d dispatches (need >= 2)" % (marker, len(marks)))
    n = len(rows)
    per = n - marks[-1]
    last = rows[marks[-1]:]
    prev = rows[marks[-2]:marks[-1]]
    t0 = last[0][0]
    span = last[-1][1] - t0
    busy = 0
    cur_s, cur_e = last[0][0], last[0][1]
    for s, e, _ in last[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = defaultdict(lambda: [0, 0])
    for s, e, nm in last:
        tot[short(nm)][0] += e - s
        tot[short(nm)][1] += 1
    # span between consecutive repetitions (start to start) = the real period incl. host gaps
    print("# dispatches per repetition: %d; period (start of previous rep -> start of last rep) %.3f ms" % (per, (t0 - prev[0][0]) / 1e6))
    print("# last repetition: span %.3f ms, busy (union) %.3f ms, idle between dispatches %.3f ms, sum of durations %.3f ms" % (
        span / 1e6, busy / 1e6, (span - busy) / 1e6, sum(e - s for s, e, _ in last) / 1e6))
    print("%-80s %6s %10s %7s" % ("kernel", "calls", "total_ms", "share"))
    for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print("%-80s %6d %10.3f %6.1f%%" % (k, c, t / 1e6, 100.0 * t / span))
    if "--list" in sys.argv:
        print("# every dispatch: start offset (us), duration (us), idle gap in front (us)")
        end_prev = last[0][0]
        for s, e, nm in last:
            print("%10.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - end_prev) / 1e3, short(nm)))
            end_prev = max(end_prev, e)


if __name__ == "__main__":
    main()
