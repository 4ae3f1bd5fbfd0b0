#!/usr/bin/env python3
"""Timeline figures from a rocprofv3 kernel trace CSV: per queue the busy time and the gaps between consecutive kernels, the time
two queues overlap, and the wall time of the window -- what `--stats` (per-kernel averages) cannot show for a multi-stream step.

    python tools/trace_timeline.py <dir with *_kernel_trace.csv> [name filter, default 'ud::lg_']
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    filt = sys.argv[2] if len(sys.argv) > 2 else "ud::lg_"
    rows = [r for r in csv.DictReader(open(path))]
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows if filt in r["Kernel_Name"]]
    ks.sort()
    # keep the second half (past warm-up)
    ks = ks[len(ks) // 2:]
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    per_q = defaultdict(list)
    for k in ks:
        per_q[k[2]].append(k)
    print("window %.1f us, %d kernels on %d queues" % ((t1 - t0) / 1e3, len(ks), len(per_q)))
    for q, lst in per_q.items():
        busy = sum(e - s for s, e, _, _ in lst)
        gaps = sorted(max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1))
        print("  queue %s: %d kernels, busy %.1f us, gaps: median %.1f us, mean %.1f us, p90 %.1f us, sum %.1f us" % (
            q, len(lst), busy / 1e3, gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(gaps) / 1e3))
    # union busy time / overlap
    ev = sorted([(s, 1) for s, e, _, _ in ks] + [(e, -1) for s, e, _, _ in ks])
    depth, last, t_any, t_two = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1: t_any += t - last
        if depth >= 2: t_two += t - last
        depth += d
        last = t
    print("  some kernel running %.1f us (%.0f %% of the window), two or more %.1f us" % (t_any / 1e3, 100.0 * t_any / (t1 - t0), t_two / 1e3))


if __name__ == "__main__":
    main()
