#!/usr/bin/env python3
"""GPU box: what happens to the trees along bench.py's drift walk -- per chunk of 100 evaluations the withheld count, the
capacity variant, node totals, row builds and forest plans.  Usage: scripts/drift_probe.py [steps=2000] [sigma=0.001]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
dev = bench.HipBackend(torch, 0, "nccl")
system = bench.load_workload("1dwc")
walk = dev.random_walk(system.pos, steps + 20, sigma, 20261004)
r = bench.Replica(dev, system, 1, steps + 20, 0, geometries=walk)
assert r.settle(20)
k = r.kernel
print("step withheld variant total_nodes max_subtree max_atoms forests builds plans pack_level ms/eval", flush=True)
for first in range(20, 20 + steps, 100):
    dev.synchronize()
    t0 = time.perf_counter()
    r.run(first, 100)
    bad = k.finish(r.stream)
    ms = 1e3 * (time.perf_counter() - t0) / 100
    try:
        row = [int(k.scalar(s)) for s in ("variant", "total_nodes", "max_subtree_nodes", "max_local_atoms", "forests", "row_builds", "pack_plans", "pack_level")]
    except Exception as exc:  # (the chunk's last evaluation was void)
        row = [str(exc)]
    print(first + 100, bad, *row, round(ms, 4), flush=True)
