#!/usr/bin/env python3
"""GPU box: the overlap-tree shape (nodes, local atoms) of every subtree of a system after the packing has settled, saved to
gpurun_out/shapes_<name>.npz -- input of the offline packing experiments (scripts/pack_sim.py).  Usage: shape_dump.py name [name ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
import bench

os.makedirs("gpurun_out", exist_ok=True)
for name in sys.argv[1:]:
    s = bench.load_workload(name)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    for step in range(6):
        f = np.zeros((s.n, 3))
        k.execute(s.jittered(step), f)
    heavy = s.ishydrogen == 0
    nodes = k.vector("subtree_nodes")[heavy].astype(np.int32)
    atoms = k.vector("subtree_atoms")[heavy].astype(np.int32)
    np.savez(f"gpurun_out/shapes_{name}.npz", nodes=nodes, atoms=atoms, forests=int(k.scalar("forests")), variant=int(k.scalar("variant")))
    print(name, "subtrees", len(nodes), "forests", int(k.scalar("forests")), "total nodes", int(nodes.sum()), "max", int(nodes.max()), int(atoms.max()))
