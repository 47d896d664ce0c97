#!/usr/bin/env python3
"""Diagnostic (GPU box, -DAGBNP_STAMPS build): per-workgroup timeline of k_tree_cavity: when every forest starts and
ends on the wall clock, its size, where it ran.  Usage:
  AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_stamps.so python scripts/wg_timeline.py [system]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
s = P.load_system(name)
ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
lib = _lib.load()
for k in range(4):
    ctx.setPositions(s.jittered(k)); ctx.getState()
nf = int(ctx.kernel.scalar("forests"))
ctx.setPositions(s.jittered(5)); ctx.getState()
buf = (C.c_ulonglong * (24 * 8192))()
lib.agbnp_debug_wg_log(buf, 8192)
log = np.array(list(buf), dtype=np.float64).reshape(8192, 24)[:nf]
t0 = log[:, 3].min()
start, end = (log[:, 3] - t0) / 100.0, (log[:, 4] - t0) / 100.0  # us (100 MHz wall clock)
life = end - start
print(f"{name}: {nf} forests; first start 0, last start {start.max():.1f} us, last end {end.max():.1f} us; mean life {life.mean():.1f} us, max {life.max():.1f}")
print("clock estimate (cycles / wall):", np.median(log[:, 5] / np.maximum(life, 1e-3)) / 1e3, "GHz")
order = np.argsort(-end)
print("last to finish:  slot roots nodes atoms start end life xcc")
for i in order[:15]:
    print(f"   {i:5d} {int(log[i,0]):3d} {int(log[i,1]):5d} {int(log[i,2]):4d} {start[i]:7.1f} {end[i]:7.1f} {life[i]:7.1f}  {int(log[i,6])&0xf}")
A = np.stack([np.ones(nf), log[:, 1], log[:, 2], log[:, 0]], 1)
coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print("life ~ %.1f + %.4f nodes + %.3f atoms + %.2f roots  (us); residual rms %.1f" % (*coef, np.sqrt(np.mean((A @ coef - life) ** 2))))
for lo, hi in ((0, 100), (100, 200), (200, 300), (300, 1000)):
    sel = (log[:, 1] >= lo) & (log[:, 1] < hi)
    if sel.any():
        print(f"  nodes {lo:3d}-{hi:4d}: {sel.sum():4d} forests, life mean {life[sel].mean():6.1f} max {life[sel].max():6.1f}, start mean {start[sel].mean():5.1f}")
cu = (log[:, 7].astype(np.int64) >> 8) & 0xf
se = (log[:, 7].astype(np.int64) >> 13) & 0x7
xcc = log[:, 6].astype(np.int64) & 0xf
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, cu) keys:", len(u), "forests per key: min", cnt.min(), "max", cnt.max())
print("xcc of slots 0..23:", [int(x) for x in xcc[:24]])
for x in range(8):
    sel = xcc == x
    if sel.any():
        print(f"  xcc {x}: {sel.sum():4d} forests, nodes {int(log[sel,1].sum()):6d}, life mean {life[sel].mean():5.1f} max {life[sel].max():5.1f}, last end {end[sel].max():5.1f}, clock {np.median(log[sel,5]/np.maximum(life[sel],1e-3))/1e3:.3f} GHz")
# per CU: forests, nodes, last end
rows = []
for k in u:
    sel = key == k
    rows.append((end[sel].max(), int(sel.sum()), int(log[sel, 1].sum()), int(log[sel, 2].sum()), k))
rows.sort(reverse=True)
print("slowest CUs (last end, forests, nodes, atoms, key):", rows[:8])
print("fastest CUs:", rows[-5:])
ends = np.array([r[0] for r in rows]); nodes_cu = np.array([r[2] for r in rows]); atoms_cu = np.array([r[3] for r in rows])
print("corr(last end of CU, nodes on CU) = %.2f, with atoms %.2f" % (np.corrcoef(ends, nodes_cu)[0, 1], np.corrcoef(ends, atoms_cu)[0, 1]))

# per-phase cycles by XCC (phases as in scripts/stamps.py): which phases make the slow XCCs slow?
names = {8: "level-2 search", 9: "level-2 rank+create", 10: "phase0", 11: "phase1", 12: "phase2/3a", 13: "phase3", 1: "pass 1", 2: "topology out",
         3: "switch radii", 4: "pass 2", 5: "root gradient", 6: "flush", 7: "node steps", 14: "gathers"}
print("phase cycles (mean per forest) by xcc:")
print("  %-22s" % "phase" + " ".join(f"xcc{x:d}".rjust(8) for x in range(8)))
for k, nm in names.items():
    row = [log[xcc == x, 8 + k].mean() if (xcc == x).any() else 0 for x in range(8)]
    print("  %-22s" % nm + " ".join(f"{v:8.0f}" for v in row))
# which slots share a CU?  (dispatch order -> placement)
print("slots per CU (first 6 CUs by key):")
for k in u[:6]:
    sl = np.flatnonzero(key == k)
    print(f"  key {k}: slots {sl.tolist()}  nodes {log[sl,1].astype(int).tolist()}  ends {[round(float(x),1) for x in end[sl]]}")
per_cu_nodes = np.array([log[key == k, 1].sum() for k in u]); per_cu_life = np.array([life[key == k].mean() for k in u])
print("per-CU nodes: min %d mean %d max %d; corr(mean life on CU, nodes on CU) = %.2f" % (per_cu_nodes.min(), per_cu_nodes.mean(), per_cu_nodes.max(), np.corrcoef(per_cu_nodes, per_cu_life)[0, 1]))
# per-phase cost model: cycles ~ c0 + c1 nodes + c2 local atoms + c3 roots
print("per-phase fit (cycles):            const   /node   /atom   /root   (mean)")
for kph, nm in names.items():
    y = log[:, 8 + kph]
    cf, *_ = np.linalg.lstsq(A, y, rcond=None)
    print("  %-22s %8.0f %7.1f %7.1f %7.0f   %8.0f" % (nm, cf[0], cf[1], cf[2], cf[3], y.mean()))
# lifetime by dispatch row (slot // 256)
rows = (np.arange(nf) // 256)
for r in range(rows.max() + 1):
    sel = rows == r
    print(f"  row {r}: {sel.sum():4d} forests, nodes mean {log[sel,1].mean():6.1f}, start {start[sel].mean():4.1f}, life mean {life[sel].mean():5.1f} max {life[sel].max():5.1f}, end mean {end[sel].mean():5.1f} max {end[sel].max():5.1f}")
