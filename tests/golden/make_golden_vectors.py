#!/usr/bin/env python3
"""Generate tests/golden/golden_vectors.npz: energies, forces, self volumes and Born radii for the
bundled structures, from the PINNED CPU oracle (oracle/agbnp_oracle.cpp; pinned by tests/test_oracle_golden.py
against the reference's v0.reference / v1.reference and the survey's recorded energies).

These vectors are oracle outputs, not outputs of the reference binary: the reference's path cannot be
built in this image (it needs OpenMM headers).  They freeze the oracle so that later
edits to it cannot drift unnoticed, and they give the GPU tests a committed per-atom target."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import openmm_agbnp_plugin_amd.systems as systems  # noqa: E402
from oracle import Oracle  # noqa: E402

out = {}
for name in ["fixture264", "trpcage", "1dwc"]:
    s = systems.load_system(name)
    for v in (0, 1):
        o = Oracle(*s.params(), version=v)
        e, f = o.execute(s.pos)
        out[f"{name}_v{v}_energy"] = np.array(e)
        out[f"{name}_v{v}_forces"] = f
        out[f"{name}_v{v}_selfvol_vdw"] = o.vector("selfvol_vdw")
        if v == 1:
            out[f"{name}_v{v}_born"] = o.vector("born")
        print(name, v, e)
np.savez_compressed(os.path.join(HERE, "golden_vectors.npz"), **out)
