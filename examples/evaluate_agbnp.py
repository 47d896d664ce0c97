#!/usr/bin/env python3
"""Python-3 counterpart of the force-evaluation part of the reference's example scripts
(example/test_agbnp.py, example/1dwc_benchmark.py): build an AGBNPForce from a structure, evaluate the
AGBNP1 (or GaussVol) energy and forces on the MI355X and time repeated evaluations.

The reference scripts run a full OpenMM MD loop (DesmondDMSFile.createSystem(implicitSolvent='AGBNP'),
LangevinIntegrator, 10 000 steps); OpenMM is outside this repository's scope, so only the AGBNP force
itself is evaluated here, and ns/day is the AGBNP-force-limited figure at the scripts' 1 fs step.

  python examples/evaluate_agbnp.py 1dwc                # bundled structure (openmm_agbnp_plugin_amd/data/1dwc.dat)
  python examples/evaluate_agbnp.py /path/to/file.dms   # Desmond .dms with an agbnp2 table
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd.AGBNPplugin import AGBNPForce


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "1dwc"
    version = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    system = P.load_dms(name) if name.endswith(".dms") else P.load_system(name)

    force = AGBNPForce()
    force.setNonbondedMethod(AGBNPForce.CutoffNonPeriodic)  # as in example/1dwc_benchmark.py:10 (inert here, as on the Reference platform)
    force.setCutoffDistance(1.0)
    force.setVersion(version)
    for r, g, a, q, h in zip(*system.params()):
        force.addParticle(r, g, a, q, bool(h))

    context = P.AGBNPContext(force)
    context.setPositions(system.pos)
    energy, forces = context.getState()
    print(f"{system.name}: {system.n} atoms ({system.nheavy} heavy), AGBNP version {version}")
    print(f"potential energy {energy:.6f} kJ/mol   max |F| {np.abs(forces).max():.3f} kJ/mol/nm   |sum F| {np.abs(forces.sum(0)).max():.2e}")

    steps = 200
    start = time.perf_counter()
    for step in range(steps):
        context.setPositions(system.jittered(step))
        context.getState()
    elapsed = time.perf_counter() - start
    ms = 1e3 * elapsed / steps
    print(f"elapsed time={elapsed:.3f}s for {steps} evaluations through the host-buffer API: {ms:.3f} ms/eval "
          f"-> {86.4 / ms:.1f} ns/day at 1 fs if AGBNP were the only cost")


if __name__ == "__main__":  # (importing the script -- a test collector, say -- runs nothing)
    main()
