"""Top-level module `AGBNPplugin`, the name of the reference's SWIG module (python/AGBNPPlugin.i:1), so that the
reference's user scripts keep their import line:

    from AGBNPplugin import AGBNPForce

Everything lives in openmm_agbnp_plugin_amd.AGBNPplugin (the gfx950 engine's host mirror); this file only re-exports it."""
from openmm_agbnp_plugin_amd.AGBNPplugin import *  # noqa: F401,F403
from openmm_agbnp_plugin_amd.AGBNPplugin import AGBNPContext, AGBNPForce, HipCalcAGBNPForceKernel, OpenMMException  # noqa: F401
