#!/usr/bin/env python3
"""GPU box: ms per evaluation through agbnp_hip_execute_host (the CPU-platform convention: host positions in, forces added
to a host array, energy returned; PCIe and synchronisation inclusive) next to the device-resident path.
Usage: python scripts/host_entry_timing.py [system] [evaluations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s = P.load_system(name)
k = P.HipCalcAGBNPForceKernel()
k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
geoms = [s.jittered(i) for i in range(32)]
f = np.zeros((s.n, 3))
for i in range(40):
    k.execute(geoms[i % 32], f)
t0 = time.perf_counter()
for i in range(reps):
    k.execute(geoms[i % 32], f)
dt = (time.perf_counter() - t0) / reps
print(f"{name}: agbnp_hip_execute_host {1e3 * dt:.4f} ms per evaluation")
from openmm_agbnp_plugin_amd import _lib
from openmm_agbnp_plugin_amd.AGBNPplugin import _dp, _ip
r, g, a, q, h = P.AGBNPForce.from_arrays(*s.params(), version=1)._arrays()
lib = _lib.load()
t0 = time.perf_counter()
for i in range(200):
    lib.agbnp_hip_update_parameters(k._h, len(r), _dp(r), _dp(g), _dp(a), _dp(q), _ip(h))
print(f"{name}: agbnp_hip_update_parameters {1e3 * (time.perf_counter() - t0) / 200:.4f} ms per call (C ABI, arrays at hand)")
