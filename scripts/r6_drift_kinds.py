#!/usr/bin/env python3
"""GPU box: bench.py's drift record (2000 evaluations along a cumulative random walk, chunks of 500 between two reads of the
overflow log), with the KINDS of what was withheld in every chunk and repeat (scalar 15) and the healed forests (scalar 17).
Usage: r6_drift_kinds.py [steps] [sigma] [chunk]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 500
warmup = 20
dev = bench.HipBackend(torch, 0, "nccl")
system = bench.load_workload("1dwc")
walk = dev.random_walk(system.pos, steps + warmup, sigma, 20261004)
r = bench.Replica(dev, system, 1, steps + warmup, 0, geometries=walk)
k = r.kernel
print("settle", r.settle(warmup), r.report, flush=True)
for first in range(warmup, warmup + steps, chunk):
    todo = list(range(first, min(first + chunk, warmup + steps)))
    for attempt in range(8):
        for step in todo:
            r.run(step, 1)
        bad = k.finish(r.stream)
        named = list(k.withheld())
        print(f"chunk at {first} attempt {attempt}: queued {len(todo)} withheld {bad} at {named[:12]} kinds {hex(int(k.scalar('overflow_kinds')))} healed {int(k.scalar('healed_forests'))} "
              f"variant {int(k.scalar('variant'))} level {int(k.scalar('pack_level'))} forests {int(k.scalar('forests'))} max nodes {int(k.scalar('max_subtree_nodes'))} "
              f"max atoms {int(k.scalar('max_local_atoms'))} row builds {int(k.scalar('row_builds'))} launches {int(k.scalar('launches'))}", flush=True)
        if not bad:
            break
        todo = [todo[i] for i in named]
