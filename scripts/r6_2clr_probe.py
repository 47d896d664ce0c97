#!/usr/bin/env python3
"""GPU box: why `bench: tree capacity did not settle (2clr)` (BENCH_r05).  Replays the secondary entry's protocol OF ROUND 5 on
2clr -- settle (8 tries of 20), a pre-heat, then the timed pass (3 tries over the SAME 200 geometries, no recovery in between) --
in fresh contexts, once per pre-heat COUNT (round 5 pre-heated by time: the count, hence the phase of the packing's 16-evaluation
plan period in which the timed pass starts, differed from box to box), and logs for every finish(): withheld count, the indices,
overflow kinds, packing level, forests.  One run answers: which phase gives up, how often, and whether try 2 and try 3 of a
failed pass fail at the same evaluation (the deterministic-replay hypothesis of VERDICT r05).
Usage: r6_2clr_probe.py [system] [first count] [last count] [step]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import openmm_agbnp_plugin_amd as P

name = sys.argv[1] if len(sys.argv) > 1 else "2clr"
c0, c1, cs = (int(sys.argv[k]) if len(sys.argv) > k else d for k, d in ((2, 0), (3, 700), (4, 20)))
steps, warmup = 200, 20
system = bench.load_workload(name)
dev = torch.device("cuda:0")
geoms = np.stack([system.jittered(7000 + s) for s in range(steps + warmup)])  # (the secondary entry's seeds)
pos = torch.tensor(geoms, dtype=torch.float64, device=dev).contiguous()
st = torch.cuda.current_stream().cuda_stream


def state(k):
    return (f"kinds {int(k.scalar('overflow_kinds'))} level {int(k.scalar('pack_level'))} forests {int(k.scalar('forests'))} "
            f"plans {int(k.scalar('pack_plans'))} variant {int(k.scalar('variant'))}")


failed = 0
for count in range(c0, c1 + 1, cs):
    k = P.HipCalcAGBNPForceKernel()
    f = P.AGBNPForce.from_arrays(*system.params(), version=1)
    f.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    k.initialize(f)
    frc = torch.zeros((system.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)

    def run(first, n):
        for s in range(first, first + n):
            k.execute_device(pos[s].data_ptr(), frc.data_ptr(), ene.data_ptr(), st)

    log = []
    settled = False
    for t in range(8):
        run(0, warmup)
        bad = k.finish(st)
        log.append(f"settle {t + 1}: withheld {bad} {list(k.withheld())[:6]} {state(k)}")
        if not bad:
            settled = True
            break
    done = 0
    while done < count:
        run(0, warmup)
        torch.cuda.synchronize()
        done += warmup
    bad = k.finish(st)
    log.append(f"preheat {done}: withheld {bad} {list(k.withheld())[:6]} {state(k)}")
    ok = False
    for t in range(3):
        run(warmup, steps)
        bad = k.finish(st)
        log.append(f"timed {t + 1}: withheld {bad} {list(k.withheld())[:6]} {state(k)}")
        if not bad:
            ok = True
            break
    verdict = "ok" if (settled and ok) else "DID NOT SETTLE"
    failed += 0 if (settled and ok) else 1
    print(f"== preheat count {count}: {verdict}")
    for ln in log:
        print("   ", ln)
    sys.stdout.flush()
    k.release()
print(f"# {failed} of {len(range(c0, c1 + 1, cs))} protocol runs did not settle")
