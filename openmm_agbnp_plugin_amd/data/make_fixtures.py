#!/usr/bin/env python3
"""Export the reference's bundled structures as package DATA (inputs only; what load_system() and the examples read).

Run in the build container only (needs /root/reference); the GPU box never runs this.

Sources (data files, not code):
  * platforms/reference/tests/gaussvol.dat  -> fixture264.dat   (the reference test's own input, 264 atoms)
  * platforms/opencl/tests/gaussvol.dat     -> fixture264_ocl.dat (same atoms, radii +0.5 A convention)
  * example/{trpcage,1dwc,2clr}_agbnp1.dms  -> {trpcage,1dwc,2clr}.dat, same 8-column layout the
    reference's test program reads from stdin
    (platforms/reference/tests/TestReferenceAGBNPForce.cpp:57 `id x y z radius charge gamma ishydrogen`),
    produced with the read-only query of SURVEY.md App. B:
        SELECT p.id,p.x,p.y,p.z,a.radius,p.charge,a.igamma,p.anum
        FROM particle p JOIN agbnp2 a ON p.id=a.id ORDER BY p.id ;  ishydrogen = (anum == 1)

Units in the .dat files are the test program's: Angstrom, e, kcal/mol/A^2.  The conversion to the
AGBNPForce::addParticle units happens in openmm_agbnp_plugin_amd/systems.py, following
TestReferenceAGBNPForce.cpp:47-70.
"""
import os
import shutil
import sqlite3

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def export_dms(name, out):
    path = f"{REF}/example/{name}"
    con = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    rows = con.execute(
        "SELECT p.id,p.x,p.y,p.z,a.radius,p.charge,a.igamma,p.anum "
        "FROM particle p JOIN agbnp2 a ON p.id=a.id ORDER BY p.id"
    ).fetchall()
    con.close()
    with open(os.path.join(HERE, out), "w") as f:
        f.write(f"{len(rows)}\n")
        for (i, x, y, z, r, q, g, anum) in rows:
            # repr() keeps every bit of the stored doubles
            f.write(f"{i} {x!r} {y!r} {z!r} {r!r} {q!r} {g!r} {1 if anum == 1 else 0}\n")
    print(out, len(rows))


if __name__ == "__main__":
    shutil.copyfile(f"{REF}/platforms/reference/tests/gaussvol.dat", os.path.join(HERE, "fixture264.dat"))
    shutil.copyfile(f"{REF}/platforms/opencl/tests/gaussvol.dat", os.path.join(HERE, "fixture264_ocl.dat"))
    export_dms("trpcage_agbnp1.dms", "trpcage.dat")
    export_dms("1dwc_agbnp1.dms", "1dwc.dat")
    export_dms("2clr_agbnp1.dms", "2clr.dat")
