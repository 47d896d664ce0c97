#!/usr/bin/env python3
"""GPU box: how the forest packing of a system behaves over a run of jittered evaluations on the device-resident path: per chunk
of evaluations the forests, the packing level (how far the assumed capacity is tightened), the plans and the withheld
evaluations with their kinds.  Usage: pack_probe.py name [chunks] [chunk length]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import openmm_agbnp_plugin_amd as P
import bench

name = sys.argv[1]
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
length = int(sys.argv[3]) if len(sys.argv) > 3 else 40
s = bench.load_workload(name)
k = P.HipCalcAGBNPForceKernel()
k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
dev = torch.device("cuda:0")
geoms = np.stack([s.jittered(9000 + i) for i in range(length)])
pos = torch.tensor(geoms, dtype=torch.float64, device=dev).contiguous()
frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
ene = torch.zeros((1,), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for c in range(chunks):
    for i in range(length):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), st)
    bad = k.finish(st)
    print(f"chunk {c}: withheld {bad} {list(k.withheld())[:6]} kinds {int(k.scalar('overflow_kinds'))} forests {int(k.scalar('forests'))} level {int(k.scalar('pack_level'))} "
          f"plans {int(k.scalar('pack_plans'))} variant {int(k.scalar('variant'))} max nodes {int(k.scalar('max_subtree_nodes'))} total {int(k.scalar('total_nodes'))}", flush=True)
