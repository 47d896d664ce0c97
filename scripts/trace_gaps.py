#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of the timed bench evaluations from a rocprofv3 --kernel-trace
database (rocpd sqlite).  Usage: python scripts/trace_gaps.py gpurun_out/prof/run_results.db [first_eval last_eval]"""
import collections, re, sqlite3, sys
import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (40, 200)


def short(n):
    m = re.search(r"(k_\w+)", n)
    s = m.group(1) if m else n[:32]
    return s


evals, cur = [], None
for n, s, e, st in rows:
    k = short(n)
    if k == "k_prep":
        if cur:
            evals.append(cur)
        cur = []
    if cur is not None:
        cur.append((k, s, e, st))
evals = evals[lo:hi]
dur, gap, span = collections.defaultdict(list), collections.defaultdict(list), []
for ev in evals:
    main = ev
    for i, (k, s, e, st) in enumerate(main):
        dur[k].append((e - s) / 1e3)
        if i:
            gap[main[i - 1][0] + " -> " + k].append((s - main[i - 1][2]) / 1e3)
    span.append((main[-1][2] - main[0][1]) / 1e3)
print("evaluations %d..%d of %d" % (lo, hi, len(evals) + lo))
for k, v in dur.items():
    print("  %-16s %7.2f us" % (k, np.mean(v)))
print("  sum %.2f us" % sum(np.mean(v) for k, v in dur.items()))
for k, v in gap.items():
    print("  gap %-40s %7.2f us" % (k, np.mean(v)))
print("  first start -> last end: %.2f us;  eval period %.2f us" % (np.mean(span), np.mean(np.diff([ev[0][1] for ev in evals])) / 1e3))
