"""GPU: the py3 counterparts of the reference's example scripts (example/test_agbnp.py, example/1dwc_benchmark.py) and
the energy-conservation check the reference does by eye (example/test_agbnp.py:55-75), asserted."""
import os
import subprocess
import sys

import numpy as np
import pytest

import openmm_agbnp_plugin_amd as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nve_energy_conservation_trpcage(gpu_required, systems):
    """Velocity Verlet at 1 fs on the device, AGBNP1 + tethers, 3000 graph-replayed steps from a 300 K start: if the
    forces were not the gradient of the energy (or an evaluation went missing) the total energy would drift
    systematically; a symplectic integrator on consistent forces only shows a bounded fluctuation."""
    pytest.importorskip("torch")
    from openmm_agbnp_plugin_amd.md import DeviceMD
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    md = DeviceMD(s, k, k_tether=2.0e4, dt=0.001, temperature=300.0, seed=3)
    md.settle()
    md.forces()
    assert k.finish() == 0
    md.run(500, "langevin", check_every=500)  # leave the perfectly tethered start: atoms now sit on AGBNP force slopes
    nve = DeviceMD(s, k, k_tether=2.0e4, dt=0.001)
    nve.x.copy_(md.x)
    nve.v.copy_(md.v)
    nve.forces()
    assert k.finish() == 0
    missed = nve.run(3000, "verlet", check_every=1000)
    assert missed == 0
    pot, kin = nve.energies()
    assert len(pot) == 3000
    total = pot + kin
    ke = kin.mean()
    assert ke > 300.0  # ~1.5 N kT = 1017 kJ/mol at 300 K, half of it after equipartition with the tethers
    q = len(total) // 4
    drift = abs(total[-q:].mean() - total[:q].mean())
    assert np.abs(total - total[0]).max() < 0.02 * ke, "total energy fluctuates by more than 2 % of the kinetic energy"
    assert drift < 0.003 * ke, f"total energy drifts: {drift:.3f} kJ/mol over 3 ps"
    # the AGBNP term matters in this balance: its energy varies by far more than the conservation error
    assert np.ptp(pot) > 10 * np.abs(total - total[0]).max()


def test_fused_integrator_steps_are_the_torch_steps(gpu_required, systems):
    """The two-launch integrator (csrc/md_kernels.hip) against the same step written in torch operations: velocity Verlet
    is deterministic, so positions, velocities and the logged energies of twenty steps must agree to rounding."""
    pytest.importorskip("torch")
    from openmm_agbnp_plugin_amd.md import DeviceMD
    s = systems("trpcage")
    runs = []
    for fused in (True, False):
        k = P.HipCalcAGBNPForceKernel()
        k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
        md = DeviceMD(s, k, k_tether=2.0e4, dt=0.001, temperature=300.0, seed=5, fused=fused)
        md.settle()
        md.x.add_(0.002 * md.torch.sin(md.x * 37.0))  # off the tether minimum
        md.forces()
        assert k.finish() == 0
        e0 = float(md.ene)
        assert md.run(20, "verlet", check_every=20) == 0
        pot, kin = md.energies()
        runs.append((e0, md.x.cpu().numpy(), md.v.cpu().numpy(), pot, kin))
    (e0a, xa, va, pa, ka), (e0b, xb, vb, pb, kb) = runs
    assert abs(e0a - e0b) < 1e-8 * abs(e0b)
    assert np.abs(xa - xb).max() < 1e-11 and np.abs(va - vb).max() < 1e-9
    assert np.abs(pa - pb).max() < 1e-7 and np.abs(ka - kb).max() < 1e-7


def test_fused_langevin_holds_the_temperature(gpu_required, systems):
    """The Philox / Box-Muller deviates of the fused Langevin step through the dynamics: 2 ps at 10 / ps friction from a
    cold start must arrive at the bath's 300 K (816 degrees of freedom: 5 % instantaneous fluctuation, less on average)."""
    pytest.importorskip("torch")
    from openmm_agbnp_plugin_amd.md import DeviceMD, KB
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    md = DeviceMD(s, k, k_tether=2.0e4, dt=0.001, temperature=300.0, friction=10.0, seed=11)
    md.settle()
    md.v.zero_()
    md.forces()
    assert k.finish() == 0
    assert md.run(2000, "langevin", check_every=1000) == 0
    _, kin = md.energies(last=800)
    temp = 2.0 * kin.mean() / (3 * s.n * KB)
    assert 280.0 < temp < 320.0, f"temperature {temp:.1f} K"


@pytest.mark.parametrize("script,args,expect", [
    ("examples/test_agbnp.py", ["trpcage", "1000", "200"], "Test energy conservation ..."),
    ("examples/1dwc_benchmark.py", ["1dwc", "1000"], "ns/day"),
    ("examples/evaluate_agbnp.py", [], None),
])
def test_example_scripts_run(gpu_required, script, args, expect):
    out = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    if expect:
        assert expect in out.stdout
    assert "WARNING" not in out.stdout
