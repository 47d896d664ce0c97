"""MI355X (gfx950) AGBNP / GaussVol implicit-solvent force path behind the reference's AGBNPForce API.

Product code only: HIP kernels + C ABI (csrc/, include/agbnp_hip.h) and the host-side mirror of the
reference's plugin interface (AGBNPplugin.py).  Nothing in this package imports the CPU oracle."""
from .AGBNPplugin import AGBNPContext, AGBNPForce, HipCalcAGBNPForceKernel, OpenMMException, host_tables  # noqa: F401
from .systems import AGBNPSystem, lattice, load_dms, load_system  # noqa: F401
