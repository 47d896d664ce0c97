#!/usr/bin/env python3
"""Diagnostic (GPU box, -DAGBNP_PAIR_STAMPS build): per-workgroup timeline of the two row-form kernels (k_rows) of one
evaluation.  Usage: AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_pstamps.so python scripts/rows_timeline.py [system]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
mode = sys.argv[2] if len(sys.argv) > 2 else "reference"
s = P.load_system(name)
os.environ.setdefault("AGBNP_HIP_ROWS", "1")
force = P.AGBNPForce.from_arrays(*s.params(), version=1)
force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
force.setCutoffDistance(1.0)
ctx = P.AGBNPContext(force)
ctx.kernel.set_mode(mode)
lib = _lib.load()
for k in range(6):
    ctx.setPositions(s.jittered(k)); ctx.getState()
SLOTS = 4096
buf = (C.c_ulonglong * (3 * SLOTS * 12))()
lib.agbnp_debug_pair_log(buf)
log = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(3, SLOTS, 12)
for kern, nm in ((0, "k_rows<born>"), (1, "k_rows<gb>"), (2, "k_rows<chain>")):
    L = log[kern]
    if kern == 1 and mode != "fast":
        continue
    if kern == 1:
        print(f"   (bookkeeping workgroup of the GB launch: entry {0.0:.2f}, end {(L[0, 3] - L[0, 0]) / 100.0:.2f} us after its entry)")
    ok = (L[:, 0] > 0) & (L[:, 3] >= L[:, 0])
    latest = L[ok, 0].max()
    ok &= L[:, 0] > latest - 20000
    W = L[ok]
    t0 = W[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    print(f"== {nm}: {int(ok.sum())} working workgroups; entries from 0 to {us(W[:, 0]).max():.2f} us; last end {us(W[:, 3]).max():.2f} us")
    phases = (("entry -> row atoms", W[:, 7] - W[:, 0]), ("-> slices in LDS (barrier)", W[:, 8] - W[:, 7]), ("-> first records", W[:, 1] - W[:, 8]),
              ("loop", W[:, 2] - W[:, 1]), ("sums -> atomics", W[:, 3] - W[:, 2]), ("lifetime", W[:, 3] - W[:, 0]))
    for pn, v in phases:
        v = v / 100.0
        print(f"   {pn:28s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    for pn, v in (("  entry -> item count (scalar)", W[:, 10] - W[:, 0]), ("  -> item word", W[:, 11] - W[:, 10]), ("  -> atoms, lists, table", W[:, 7] - W[:, 11])):
        v = v / 100.0
        print(f"   {pn:28s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    hw = W[:, 4].astype(np.int64); xcc = (W[:, 5].astype(np.int64)) & 15
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    life = (W[:, 3] - W[:, 0]) / 100.0
    pro = (W[:, 8] - W[:, 0]) / 100.0
    uniq, inv, cnt = np.unique(cuid, return_inverse=True, return_counts=True)
    per = cnt[inv]
    print(f"   CUs used {len(uniq)}; workgroups per CU: " + ", ".join(f"{k}: {int((cnt == k).sum())} CUs" for k in sorted(set(cnt))))
    for k in sorted(set(cnt)):
        print(f"     workgroups on CUs that hold {k}: lifetime mean {life[per == k].mean():.2f} max {life[per == k].max():.2f}; prologue mean {pro[per == k].mean():.2f} max {pro[per == k].max():.2f}")
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print(f"     XCD {x}: {int(sel.sum())} workgroups, lifetime mean {life[sel].mean():.2f} p90 {np.percentile(life[sel], 90):.2f} max {life[sel].max():.2f}; prologue mean {pro[sel].mean():.2f}; last end {us(W[sel, 3]).max():.2f}")
    steps = W[:, 9]
    loop = (W[:, 2] - W[:, 1]) / 100.0
    print(f"   entries per part: mean {W[:, 6].mean():.0f} max {W[:, 6].max():.0f}; steps mean {steps.mean():.1f}; us per step {np.median(loop / np.maximum(steps, 1)):.3f}")
    end = us(W[:, 3])
    ent = us(W[:, 0])
    idx = np.flatnonzero(ok)
    late = ent > 3.0
    print(f"   late entrants (> 3 us): {int(late.sum())}: block numbers {idx[late][:24].tolist()} of {int(idx.max()) + 1} stamped blocks; "
          f"working blocks by index range: <260 {int((idx < 260).sum())}, <520 {int((idx < 520).sum())}, <780 {int((idx < 780).sum())}, <1040 {int((idx < 1040).sum())}")
    print(f"   entries: p50 {np.median(ent):.2f} p90 {np.percentile(ent, 90):.2f} p99 {np.percentile(ent, 99):.2f} max {ent.max():.2f}")
    print(f"   ends: p10 {np.percentile(end, 10):.2f} p50 {np.median(end):.2f} p90 {np.percentile(end, 90):.2f} max {end.max():.2f}")
