#!/usr/bin/env python3
"""Python-3 counterpart of the reference's example/1dwc_benchmark.py on the MI355X engine: a device-resident
Langevin loop (300 K, 1/ps friction, 1 fs step, as example/1dwc_benchmark.py:20) whose only force field term is
AGBNP1 -- the reference gets its bonded and Coulomb/LJ terms from OpenMM's OPLS system, which is outside this
repository; atoms are tethered harmonically to their start positions instead so that the geometry stays a protein.
One MD step = integrator half kicks + one agbnp_hip_execute_device, captured once as a HIP graph and replayed;
prints the state every 1000 steps like the reference's StateDataReporter and the elapsed time / ns/day at the end.

  python examples/1dwc_benchmark.py [system=1dwc] [steps=10000]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd.AGBNPplugin import AGBNPForce, HipCalcAGBNPForceKernel

name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
system = P.load_dms(name) if name.endswith(".dms") else P.load_system(name)

force = AGBNPForce()
force.setNonbondedMethod(AGBNPForce.CutoffNonPeriodic)  # example/1dwc_benchmark.py:10
force.setCutoffDistance(1.0)
force.setVersion(1)
for r, g, a, q, h in zip(*system.params()):
    force.addParticle(r, g, a, q, bool(h))
kernel = HipCalcAGBNPForceKernel()
kernel.initialize(force)

dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
kB, T, gamma, dt = 0.0083144626, 300.0, 1.0, 0.001        # kJ/mol/K, K, 1/ps, ps
mass = torch.tensor(np.where(system.ishydrogen == 1, 1.008, 12.0)[:, None], **f64)  # amu (heavy atoms: carbon-like)
k_tether = 1.0e5                                             # kJ/mol/nm^2 (bond-like: positions fluctuate by ~0.005 nm)
x0 = torch.tensor(system.pos, **f64)
x = x0.clone()
v = torch.randn_like(x) * torch.sqrt(kB * T / mass)
frc = torch.zeros_like(x)
ene = torch.zeros(1, **f64)
c1 = float(np.exp(-gamma * dt))
c2 = torch.sqrt((1.0 - c1 * c1) * kB * T / mass)
noise = torch.empty_like(x)


def forces():
    frc.copy_(-k_tether * (x - x0))
    ene.zero_()
    kernel.execute_device(x.data_ptr(), frc.data_ptr(), ene.data_ptr(), torch.cuda.current_stream().cuda_stream)


def md_step():  # BAOAB
    v.add_(frc / mass, alpha=0.5 * dt)
    x.add_(v, alpha=0.5 * dt)
    noise.normal_()
    v.mul_(c1).add_(c2 * noise)
    x.add_(v, alpha=0.5 * dt)
    forces()
    v.add_(frc / mass, alpha=0.5 * dt)


side = torch.cuda.Stream()
with torch.cuda.stream(side):  # warm-up outside the capture (allocations, capacity negotiation, forest packing)
    forces()
    for _ in range(5):
        md_step()
    assert kernel.finish(side.cuda_stream) == 0
torch.cuda.synchronize()


def capture():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        md_step()
    return g, kernel.generation()


graph, generation = capture()

print(f"{system.name}: {system.n} atoms, AGBNP1 + tethers, Langevin 300 K, 1 fs, engine on {torch.cuda.get_device_name(0)}")
print('#"Step","AGBNP Energy (kJ/mole)","Temperature (K)"')
start = time.perf_counter()
for step in range(1, nsteps + 1):
    graph.replay()
    if step % 1000 == 0:
        # the device logs every replayed step whose trees outgrew their store: such a step got NO AGBNP force (its
        # outputs are withheld, never partial).  A production driver would roll back to its last checkpoint here.
        missed = kernel.finish(torch.cuda.current_stream().cuda_stream)
        if missed:
            first = step - 1000 + kernel.withheld()[0] + 1
            print(f"  ({missed} step(s) from step {first} on ran without the AGBNP term: tree capacity exceeded)")
        if kernel.generation() != generation:  # the capacity variant was raised: the captured kernels are stale
            graph, generation = capture()
        ke = 0.5 * float((mass * v * v).sum())
        print(f"{step},{float(ene):.4f},{2.0 * ke / (3 * system.n * kB):.2f}")
torch.cuda.synchronize()
elapsed = time.perf_counter() - start
print(f"elapsed time={elapsed:.3f}s   {1e3 * elapsed / nsteps:.4f} ms/step   {86.4 * nsteps / (elapsed * 1e3):.1f} ns/day (1 fs steps)")
