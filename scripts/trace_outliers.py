#!/usr/bin/env python3
"""Names the launches behind the maxima of a rocprofv3 kernel summary (`k_rows<0>` max 93 us,
`k_tree_cavity_five` max 73.7 us, `k_tree_pseudo` max 43.6 us in profiles/r05/bench_1dwc_kernel_stats.csv -- which launches are
those?).  Reads the kernel trace of the SAME run (rocprofv3 --kernel-trace --output-format csv: *_kernel_trace.csv), numbers
the evaluations (one k_tree_cavity* launch each), and prints for every kernel its longest launches with the evaluation they
belong to, what else ran in that evaluation (k_masks = a fresh context's first neighbour masks), how that evaluation's other
launches compare with their medians (a Born-rows launch far above its median = it rebuilt the neighbour rows, a little above =
it carried the mask-renewal tiles), and where in the process the evaluation sits (context number, evaluation of the context).
Usage: trace_outliers.py <kernel_trace.csv> [top]"""
import collections
import csv
import re
import sys

import numpy as np

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"


def short(n):
    m = re.search(r"(k_\w+(<[^>(]*>)?)", n)
    s = m.group(1) if m else n[:40]
    return re.sub(r"<(\d+)[^>]*>", r"<\1>", s) if s.startswith("k_rows") else re.sub(r"<.*", "", s)


launches = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r[name_key])) for r in rows), key=lambda t: t[0])
launches = [l for l in launches if l[2].startswith("k_")]
# evaluations: a k_tree_cavity* launch opens one (k_prep / k_masks in front of it belong to it)
evals, pending = [], []
for s, e, k in launches:
    if k in ("k_prep", "k_masks", "k_order_maps", "k_row_atoms"):
        pending.append((s, e, k))
        continue
    if k.startswith("k_tree_cavity"):
        evals.append(pending + [(s, e, k)])
        pending = []
    elif evals:
        evals[-1].append((s, e, k))
# contexts: a gap of more than 20 ms between two evaluations = a new context (bench.py creates one per record)
ctx, ctx_no, ev_in_ctx = [], 0, 0
for i, ev in enumerate(evals):
    if i and ev[0][0] - evals[i - 1][-1][1] > 20e6:
        ctx_no, ev_in_ctx = ctx_no + 1, 0
    ctx.append((ctx_no, ev_in_ctx))
    ev_in_ctx += 1
dur = collections.defaultdict(list)
for i, ev in enumerate(evals):
    for s, e, k in ev:
        dur[k].append(((e - s) / 1e3, i))
med = {k: float(np.median([d for d, _ in v])) for k, v in dur.items()}
print(f"# {path}: {len(launches)} launches, {len(evals)} evaluations in {ctx_no + 1} contexts")
for k in sorted(dur, key=lambda k: -max(d for d, _ in dur[k])):
    v = sorted(dur[k], reverse=True)[:top]
    print(f"{k}: {len(dur[k])} launches, median {med[k]:.2f} us, mean {np.mean([d for d, _ in dur[k]]):.2f}")
    for d, i in v:
        others = ", ".join(f"{kk} {((e - s) / 1e3):.1f} (median {med[kk]:.1f})" for s, e, kk in evals[i] if kk != k)
        print(f"    {d:8.2f} us  evaluation {i} = context {ctx[i][0]}, its evaluation {ctx[i][1]};  same evaluation: {others}")
