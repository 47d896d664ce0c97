#!/usr/bin/env python3
"""GPU box: forests / subtree statistics of a system after the packing has settled.  Usage: forest_probe.py name [name ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
import bench

for name in sys.argv[1:]:
    s = bench.load_workload(name)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    for step in range(6):
        f = np.zeros((s.n, 3))
        k.execute(s.jittered(step), f)
    nodes = k.vector("subtree_nodes")[s.ishydrogen == 0]
    atoms = k.vector("subtree_atoms")[s.ishydrogen == 0]
    print(name, "atoms", s.n, "heavy", s.nheavy, "variant", int(k.scalar("variant")), "forests", int(k.scalar("forests")), "total nodes", int(k.scalar("total_nodes")),
          "max nodes", int(nodes.max()), "mean nodes", round(nodes.mean(), 1), "mean local atoms", round(atoms.mean(), 1), "max local atoms", int(atoms.max()),
          "subtrees >= 214 nodes", int((nodes >= 214).sum()), ">= 428", int((nodes >= 428).sum()),
          "node histogram (50s)", np.histogram(nodes, bins=[0, 50, 100, 150, 200, 250, 300, 350, 400, 450, 500, 1000])[0].tolist())
