#!/usr/bin/env python3
"""Kernel trace -> fraction of the busiest stretch with no kernel running, gaps between consecutive kernels of a queue."""
import csv, glob, statistics, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
# the timed region = the longest run of k_preprocess_vec launches spaced < 2 ms apart
pre = [e for e in ev if "k_preprocess_vec" in e[2]]
best, cur = [], [pre[0]]
for a, b in zip(pre, pre[1:]):
    if b[0] - a[0] < 2_000_000: cur.append(b)
    else:
        if len(cur) > len(best): best = cur
        cur = [b]
if len(cur) > len(best): best = cur
t0, t1 = best[2][0], best[-3][0]                       # drop the ramp at both ends
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cs, ce = 0, win[0][0], win[0][1]
for s, e, _, _ in win[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
clips = len([e for e in win if "k_preprocess_vec" in e[2]])
tot = sum(e - s for s, e, _, _ in win)
print("window %.1f us, %d clips -> %.1f us per clip; some kernel running %.1f %% of it; sum of durations %.1f us per clip (avg concurrency %.2f)"
      % ((t1 - t0) / 1e3, clips, (t1 - t0) / 1e3 / clips, 100 * busy / (t1 - t0), tot / 1e3 / clips, tot / (t1 - t0)))
q = defaultdict(list)
for e in win: q[e[3]].append(e)
for k, v in sorted(q.items()):
    gaps = [b[0] - a[1] for a, b in zip(v, v[1:])]
    pos = [g for g in gaps if g > 0]
    print("queue %s: %d kernels, %d gaps > 0, median %.1f us, sum %.1f us per clip" % (k, len(v), len(pos), statistics.median(pos) / 1e3 if pos else 0, sum(pos) / 1e3 / max(1, clips)))
