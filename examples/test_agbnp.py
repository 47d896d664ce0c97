#!/usr/bin/env python3
"""Python-3 counterpart of the reference's example/test_agbnp.py on the MI355X engine: energy of the start structure,
minimisation, Langevin equilibration (300 K, 1/ps, 0.5 fs: example/test_agbnp.py:37), then the energy-conservation
check with a Verlet integrator at 1 fs (example/test_agbnp.py:55-75), printing step / potential / total energy /
temperature like the reference's StateDataReporter.  AGBNP1 + tethers (the OPLS terms of the reference's system come
from OpenMM and are outside this repository; see openmm_agbnp_plugin_amd/md.py).

  python examples/test_agbnp.py [system=trpcage] [equilibration steps=10000] [nve steps=1000]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import openmm_agbnp_plugin_amd as P
from AGBNPplugin import AGBNPForce, HipCalcAGBNPForceKernel
from openmm_agbnp_plugin_amd.md import DeviceMD


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "trpcage"
    n_equil = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n_nve = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    system = P.load_dms(name) if name.endswith(".dms") else P.load_system(name)
    print("Started at: " + str(time.asctime()))

    force = AGBNPForce()
    force.setNonbondedMethod(AGBNPForce.NoCutoff)  # example/test_agbnp.py:17
    force.setVersion(1)                            # implicitSolvent='AGBNP'
    for r, g, a, q, h in zip(*system.params()):
        force.addParticle(r, g, a, q, bool(h))
    kernel = HipCalcAGBNPForceKernel()
    kernel.initialize(force)

    md = DeviceMD(system, kernel, dt=0.0005, temperature=300.0, friction=1.0)
    md.settle()
    md.forces()
    kernel.finish()
    print(f"{float(md.ene):.4f} kJ/mol")

    print("Minimization/equilibration ...")
    md.v.zero_()
    md.run(200, "descent", check_every=200)
    md.v.copy_(md.torch.randn_like(md.v) * md.torch.sqrt(0.0083144626 * 300.0 / md.mass))
    print('#"Step","Potential Energy (kJ/mole)","Total Energy (kJ/mole)","Temperature (K)"')


    def report(m):
        pot, kin = m.energies(last=1)
        print(f"{m.steps_done},{pot[0]:.4f},{pot[0] + kin[0]:.4f},{2.0 * kin[0] / (3 * system.n * 0.0083144626):.2f}")


    missed = md.run(n_equil, "langevin", check_every=1000, on_report=report)

    print("Test energy conservation ...")
    nve = DeviceMD(system, kernel, dt=0.001)
    nve.x.copy_(md.x)
    nve.v.copy_(md.v)
    nve.x0.copy_(md.x0)
    nve.forces()
    kernel.finish()
    start = time.perf_counter()
    missed += nve.run(n_nve, "verlet", check_every=max(n_nve, 1))
    elapsed = time.perf_counter() - start
    pot, kin = nve.energies()
    for k in range(9, len(pot), 10):  # every 10 steps, like the reference's reporter
        print(f"{k + 1},{pot[k]:.4f},{pot[k] + kin[k]:.4f},{2.0 * kin[k] / (3 * system.n * 0.0083144626):.2f}")
    tot = pot + kin
    print(f"total energy: start {tot[0]:.4f}, end {tot[-1]:.4f}, max deviation {np.abs(tot - tot[0]).max():.4f} kJ/mol "
          f"(mean kinetic energy {kin.mean():.1f})")
    if missed:
        print(f"WARNING: {missed} step(s) ran without the AGBNP term (tree capacity exceeded)")
    print("elapsed time=" + str(elapsed) + "s")


if __name__ == "__main__":  # (importing the script -- a test collector, say -- runs nothing)
    main()
