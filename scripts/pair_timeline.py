#!/usr/bin/env python3
"""Diagnostic (GPU box, -DAGBNP_PAIR_STAMPS build): per-workgroup timeline of the three pair kernels of one
evaluation: when every tile starts, how long its prologue / walk / epilogue take, where it ran.  Usage:
  AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_pstamps.so python scripts/pair_timeline.py [system]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
s = P.load_system(name)
ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
lib = _lib.load()
for k in range(6):
    ctx.setPositions(s.jittered(k)); ctx.getState()
SLOTS = 4096
buf = (C.c_ulonglong * (3 * SLOTS * 12))()
lib.agbnp_debug_pair_log(buf)
log = np.frombuffer(buf, dtype=np.uint64).astype(np.float64).reshape(3, SLOTS, 12)
names = ["k_born_tiles", "k_gb_tiles", "k_dborn_tiles"]
t_first = None
prev_end = None
for kern in (0, 1, 2):
    L = log[kern]
    ran = (L[:, 0] > 0)
    worked = ran & (L[:, 3] > 0) & (L[:, 3] >= L[:, 0])  # culled workgroups leave no end stamp (or a stale one)
    # keep the last evaluation only
    if not worked.any():
        print(names[kern], "no stamps"); continue
    latest = L[worked, 0].max()
    worked &= L[:, 0] > latest - 20000  # 200 us window
    ran &= L[:, 0] > latest - 20000
    t0 = L[ran, 0].min()
    if t_first is None:
        t_first = t0
    us = lambda v: (v - t0) / 100.0
    if kern == 1:  # workgroup 0 of the GB launch is the bookkeeping role
        print(f"   bookkeeping workgroup: entry {us(L[0, 0]):.2f} us, end {us(L[0, 3]):.2f} us")
        worked[0] = False
        pass
        if L[0, 7] > 0:
            print("   bookkeeping phases (us from its entry): shapes+histogram %.2f, scan+classes %.2f, forest_start %.2f, placement %.2f" % tuple(
                (L[0, k] - L[0, 0]) / 100.0 for k in (7, 8, 9, 10)))
    W = L[worked]
    start, rec, walk, end = us(W[:, 0]), us(W[:, 1]), us(W[:, 2]), us(W[:, 3])
    print(f"== {names[kern]}: {int(ran.sum())} workgroups entered, {int(worked.sum())} worked; kernel's first entry at {(t0 - t_first) / 100.0:.2f} us"
          + (f" ({(t0 - prev_end) / 100.0:.2f} us after the previous kernel's last stamp)" if prev_end is not None else ""))
    print(f"   entry: last at {us(L[ran, 0]).max():.2f} us;  working workgroups: start median {np.median(start):.2f} max {start.max():.2f}")
    for nm, v in (("prologue (entry -> records in LDS)", rec - start), ("walk", walk - rec), ("epilogue (sums -> atomics issued)", end - walk), ("lifetime", end - start)):
        print(f"   {nm:36s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    if kern == 0 and (W[:, 7] > 0).all():
        for nm, v in (("  entry -> item known", us(W[:, 7]) - start), ("  -> records arrived", us(W[:, 8]) - us(W[:, 7])),
                      ("  -> tables in LDS", us(W[:, 9]) - us(W[:, 8])), ("  -> barrier passed", rec - us(W[:, 9]))):
            print(f"   {nm:36s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
    print(f"   last end stamp at {end.max():.2f} us; ends: p50 {np.median(end):.2f} p90 {np.percentile(end, 90):.2f}")
    hw = W[:, 4].astype(np.int64); xcc = W[:, 5].astype(np.int64) & 0xf
    key = xcc * 1000 + ((hw >> 13) & 0x7) * 16 + ((hw >> 8) & 0xf)
    u, cnt = np.unique(key, return_counts=True)
    print(f"   CUs used {len(u)}; working workgroups per CU min {cnt.min()} mean {cnt.mean():.1f} max {cnt.max()}")
    # does a CU with more workgroups finish later?
    last_by_cu = np.array([end[key == k].max() for k in u])
    for c in sorted(set(cnt.tolist())):
        sel = cnt == c
        print(f"      CUs with {c:2d} workgroups: {int(sel.sum()):3d}, last end mean {last_by_cu[sel].mean():6.2f} max {last_by_cu[sel].max():6.2f}")
    # tile kinds
    item = W[:, 6].astype(np.int64)
    I, J = item & 0xfff, (item >> 12) & 0xfff
    kinds = {"diagonal": I == J, "off-diagonal": I != J}
    if kern == 1:
        kinds = {"strip": (item & (1 << 24)) != 0, "tile": ((item & (1 << 24)) == 0) & (I != J), "diagonal": ((item & (1 << 24)) == 0) & (I == J)}
    for nm, sel in kinds.items():
        if sel.any():
            print(f"      {nm:13s} {int(sel.sum()):5d}: walk mean {np.mean((walk - rec)[sel]):6.2f} us, lifetime mean {np.mean((end - start)[sel]):6.2f}")
    # start order vs block index: is dispatch in order and how long does the ramp take
    idx = np.flatnonzero(worked)
    print("   entry time by block index (us):", " ".join(f"{int(i)}:{start[np.searchsorted(idx, i)]:.1f}" for i in idx[:: max(1, len(idx) // 12)]))
    prev_end = W[:, 3].max()
