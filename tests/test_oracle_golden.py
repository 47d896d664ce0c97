"""CPU: the oracle against every known answer the reference ships or the survey recorded (tests/pins.py)."""
import numpy as np
import pytest

from oracle import Oracle
from tests.pins import PROBE, REFERENCE_PRINTED, SURVEY_ENERGIES, SURVEY_TREE


def sig(x, n=6):
    return float(f"{x:.{n}g}")


@pytest.mark.parametrize("version", [0, 1])
def test_reference_printed_known_answers(systems, version):
    s = systems("fixture264")
    o = Oracle(*s.params(), version=version)
    e1, f = o.execute(s.pos)
    want = REFERENCE_PRINTED[version]
    assert sig(e1) == want["energy"]
    if version == 0:
        assert sig(o.scalar("e_vol1")) == want["e_vol1"]
        assert sig(o.scalar("e_vol2")) == want["e_vol2"]
    p = s.pos.copy()
    p[PROBE["atom"], PROBE["direction"]] += PROBE["offset"]
    e2, _ = o.execute(p)
    assert sig(e2) == want["energy_moved"]
    assert sig(e2 - e1) == want["change"]
    assert sig(-f[PROBE["atom"], PROBE["direction"]] * PROBE["offset"]) == want["change_from_gradient"]


@pytest.mark.parametrize("name", ["fixture264", "fixture264_ocl", "trpcage"])
def test_survey_energies_small(systems, name):
    s = systems(name)
    for version in (0, 1):
        e, f = Oracle(*s.params(), version=version).execute(s.pos)
        want, rtol = SURVEY_ENERGIES[name][version], SURVEY_ENERGIES[name][2]
        assert abs(e - want) <= rtol * abs(want)
        assert np.abs(f.sum(axis=0)).max() < 1e-9  # no net force


def test_survey_energy_thrombin_v1(systems):
    s = systems("1dwc")
    o = Oracle(*s.params(), version=1)
    e, f = o.execute(s.pos)
    want, rtol = SURVEY_ENERGIES["1dwc"][1], SURVEY_ENERGIES["1dwc"][2]
    assert abs(e - want) <= rtol * abs(want)
    counts, max_subtree, max_children = SURVEY_TREE["1dwc"]
    st = o.tree_stats()
    assert st["level_counts"][2:8] == counts and st["level_counts"][8] == 0
    assert st["max_subtree"] == max_subtree and st["max_children"] == max_children
    assert int(o.scalar("slots")) == 216146  # SURVEY.md s.8 header


@pytest.mark.parametrize("name", ["fixture264", "trpcage"])
def test_survey_tree_statistics(systems, name):
    s = systems(name)
    o = Oracle(*s.params(), version=0)
    o.execute(s.pos)
    counts, max_subtree, max_children = SURVEY_TREE[name]
    st = o.tree_stats()
    assert st["level_counts"][1] == s.n
    assert st["level_counts"][2:8] == counts
    assert st["max_subtree"] == max_subtree and st["max_children"] == max_children


@pytest.mark.parametrize("version", [0, 1])
def test_forces_are_the_gradient(systems, version):
    """Central finite differences (the recipe of TestReferenceAGBNPForce.cpp:93-128, but two-sided)."""
    s = systems("trpcage")
    o = Oracle(*s.params(), version=version)
    _, f = o.execute(s.pos)
    rng = np.random.default_rng(7)
    h = 1e-5
    for atom in rng.choice(s.n, size=6, replace=False):
        for d in range(3):
            pp, pm = s.pos.copy(), s.pos.copy()
            pp[atom, d] += h
            pm[atom, d] -= h
            fd = -(o.execute(pp)[0] - o.execute(pm)[0]) / (2 * h)
            assert abs(fd - f[atom, d]) < 2e-4 * max(1.0, abs(f[atom, d]))


def test_force_accumulation_and_energy_return(systems):
    s = systems("fixture264")
    o = Oracle(*s.params(), version=1)
    e, f = o.execute(s.pos)
    base = np.full((s.n, 3), 3.25)
    e2, f2 = o.execute(s.pos, force_accum=base)
    assert e2 == e
    np.testing.assert_allclose(f2 - base, f, rtol=0, atol=1e-10)


def test_multiple_gamma_values_rejected(systems):
    s = systems("fixture264")
    g = s.gamma.copy()
    heavy = np.flatnonzero(s.ishydrogen == 0)
    g[heavy[3]] *= 1.5
    from oracle.oracle import OracleError
    with pytest.raises(OracleError, match="multiple gamma"):
        Oracle(s.radius, g, s.alpha, s.charge, s.ishydrogen, version=1)


def test_update_parameters_semantics(systems):
    s = systems("fixture264")
    from oracle.oracle import OracleError
    o = Oracle(*s.params(), version=1)
    e0, _ = o.execute(s.pos)
    o.update(s.radius, s.gamma, s.alpha * 0.5, s.charge * 0.9, s.ishydrogen)
    e1, _ = o.execute(s.pos)
    fresh, _ = Oracle(s.radius, s.gamma, s.alpha * 0.5, s.charge * 0.9, s.ishydrogen, version=1).execute(s.pos)
    assert e1 != e0 and abs(e1 - fresh) < 1e-9
    with pytest.raises(OracleError, match="changing atomic radii"):
        o.update(s.radius + 0.01, s.gamma, s.alpha, s.charge, s.ishydrogen)
    flip = s.ishydrogen.copy()
    flip[np.flatnonzero(s.ishydrogen == 0)[0]] = 1
    with pytest.raises(OracleError, match="heavy/hydrogen"):
        o.update(s.radius, s.gamma, s.alpha, s.charge, flip)


def test_fast_mode_switch_is_tied_to_the_pinned_path():
    """The oracle's cutoff switch (restatement of the OpenCL platform's pair truncation) has no reference-held vector:
    it is tied to the pinned path by its limit -- a cutoff beyond the system's extent changes nothing -- and must
    change the numbers when it bites."""
    import openmm_agbnp_plugin_amd as P
    s = P.load_system("fixture264")
    e0, f0 = Oracle(*s.params(), version=1).execute(s.pos)
    e1, f1 = Oracle(*s.params(), version=1, cutoff=50.0).execute(s.pos)
    assert e1 == e0 and np.array_equal(f1, f0)
    e2, f2 = Oracle(*s.params(), version=1, cutoff=1.0).execute(s.pos)
    assert abs(e2 - e0) > 1.0 and np.abs(f2 - f0).max() > 1e-2
    # version 0 has no pair stage: the switch is inert
    e3, f3 = Oracle(*s.params(), version=0, cutoff=1.0).execute(s.pos)
    e4, f4 = Oracle(*s.params(), version=0).execute(s.pos)
    assert e3 == e4 and np.array_equal(f3, f4)
