#!/usr/bin/env python3
"""Developer check (GPU box): engine vs CPU oracle on the bundled systems; prints diffs and timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmm_agbnp_plugin_amd as P
from oracle import Oracle

names = sys.argv[1:] or ["fixture264", "trpcage", "1dwc"]
for name in names:
    s = P.load_system(name)
    for v in (0, 1):
        o = Oracle(*s.params(), version=v)
        t = time.time(); eo, fo = o.execute(s.pos); to = time.time() - t
        f = P.AGBNPForce.from_arrays(*s.params(), version=v)
        ctx = P.AGBNPContext(f)
        ctx.setPositions(s.pos)
        t = time.time(); e, fg = ctx.getState(); t1 = time.time() - t
        t = time.time(); e, fg = ctx.getState(); t2 = time.time() - t
        k = ctx.kernel
        print(f"{name} v{v} N={s.n}: E_gpu={e:.10f} E_cpu={eo:.10f} dE={e-eo:.3e} max|dF|={np.abs(fg-fo).max():.3e} "
              f"maxF={np.abs(fo).max():.1f} cpu={to*1e3:.1f}ms gpu(first)={t1*1e3:.1f}ms gpu(host api)={t2*1e3:.2f}ms "
              f"variant={k.scalar('variant'):.0f} maxnodes={k.scalar('max_subtree_nodes'):.0f} total={k.scalar('total_nodes'):.0f}", flush=True)
        print("   E parts gpu:", k.scalar('e_vol1'), k.scalar('e_vol2'), k.scalar('e_atom'), k.scalar('e_gb_pair'),
              " cpu:", o.scalar('e_vol1'), o.scalar('e_vol2'), o.scalar('e_gb') + o.scalar('e_vdw'))
        sv = k.vector('selfvol_vdw'); print("   max|d selfvol|", np.abs(sv - o.vector('selfvol_vdw')).max(), flush=True)
        if v == 1:
            print("   max|d born|", np.abs(k.vector('born') - o.vector('born')).max(), flush=True)
