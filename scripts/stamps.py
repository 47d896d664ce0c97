#!/usr/bin/env python3
"""Diagnostic (GPU box): per-phase shader-clock shares of k_tree_cavity from the -DAGBNP_STAMPS build.
Usage: AGBNP_HIP_LIBRARY=build/diag/libagbnp_hip_stamps.so python scripts/stamps.py [system]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
s = P.load_system(name)
ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
ctx.setPositions(s.pos); ctx.getState()
lib.agbnp_debug_stamps(buf, 1)
lib.agbnp_debug_stamps_max((C.c_ulonglong * 16)(), 1)
reps = 20
for k in range(reps):
    ctx.setPositions(s.jittered(k)); ctx.getState()
lib.agbnp_debug_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64) / reps
pseudo = os.environ.get("AGBNP_STAMPS_PSEUDO") is not None  # library built with -DAGBNP_STAMPS_PSEUDO
names = {0: "build (total)", 1: "(unused)", 2: "pass 1 + topology out", 3: "switch radii", 4: "pass 2", 5: "root gradient", 6: "flush",
         8: " build: level-2 scan", 9: " build: level-2 rank+create", 10: " build: phase0 (tasks/scan/map)", 11: " build: phase1 (volumes)",
         12: " build: phase2 (count/scan)", 13: " build: phase3 (rank+create)",
         7: " passes: node step (1+2)", 14: " passes: gather"}
if pseudo:
    names = {0: "load paths + atoms", 1: "volume pass", 2: "root gradient + flush", 7: " pass: node step (1+2)", 14: " pass: gather"}
mx = (C.c_ulonglong * 16)()
lib.agbnp_debug_stamps_max(mx, 1)
mx = np.array(list(mx), dtype=np.float64)
tot = v[:7].sum()
nfor = max(int(ctx.kernel.scalar('forests')), 1)
print(f"{name}: cycles per evaluation summed over the workgroups (lane 0), total {tot:.3e}; forests {int(ctx.kernel.scalar('forests'))}")
for k, n in names.items():
    print(f"  {n:34s} {v[k]:12.3e}  {100*v[k]/tot:5.1f}%   per subtree {v[k]/nfor:9.0f} cyc   slowest per phase {mx[k]:9.0f}")
print(f"slowest workgroup: total {mx[15]:.0f} cyc (mean {tot/nfor:.0f}); build {mx[0]:.0f}, pass1 {mx[2]:.0f}, pass2 {mx[4]:.0f}")
