"""CPU oracle (TEST INFRASTRUCTURE).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package; the product package openmm_agbnp_plugin_amd never does."""
from .oracle import Oracle, build_oracle, oracle_lib_path  # noqa: F401
