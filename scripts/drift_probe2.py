#!/usr/bin/env python3
"""GPU box: the drift walk evaluation by evaluation around the step where it starts to overflow.  Usage: drift_probe2.py first last"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = bench.HipBackend(torch, 0, "nccl")
system = bench.load_workload("1dwc")
walk = dev.random_walk(system.pos, 2020, 0.001, 20261004)
r = bench.Replica(dev, system, 1, 2020, 0, geometries=walk)
k = r.kernel
r.run(lo - 8, 8)
print("settle", k.finish(r.stream), flush=True)
for step in range(lo, hi):
    r.run(step, 1)
    bad = k.finish(r.stream)
    kinds = int(k.scalar("overflow_kinds"))
    try:
        row = [int(k.scalar(s)) for s in ("variant", "total_nodes", "max_subtree_nodes", "max_local_atoms", "forests", "pack_plans", "pack_level", "pack_age")]
    except Exception as exc:
        row = ["void"]
    if bad or step % 10 == 0:
        print(step, "withheld", bad, "kinds", hex(kinds), *row, flush=True)
    if bad:  # repeat as a driver would
        for attempt in range(6):
            r.run(step, 1)
            b2 = k.finish(r.stream)
            kinds = int(k.scalar("overflow_kinds"))
            try:
                row = [int(k.scalar(s)) for s in ("variant", "total_nodes", "max_subtree_nodes", "max_local_atoms", "forests", "pack_plans", "pack_level", "pack_age")]
            except Exception:
                row = ["void"]
            print("   repeat", attempt, "withheld", b2, "kinds", hex(kinds), *row, flush=True)
            if not b2:
                break
