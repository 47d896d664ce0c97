"""CPU: host logic of the product (AGBNPForce mirror, C-ABI surface, host-side I4 tables).  No compute calls."""
import os
import re

import numpy as np
import pytest

import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "agbnp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(agbnp_hip_[a-z_]+)\s*\(", header)))
    assert declared, "no declarations found in include/agbnp_hip.h"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"libagbnp_hip.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == declared


def test_force_defaults_match_reference():
    f = P.AGBNPForce()
    # AGBNPForce::AGBNPForce(): NoCutoff, cutoff 1.0, version 1, solvent radius 1.0*ANG (AGBNPForce.cpp:15)
    assert f.getNonbondedMethod() == P.AGBNPForce.NoCutoff == 0
    assert f.getCutoffDistance() == 1.0
    assert f.getVersion() == 1
    assert f.getSolventRadius() == pytest.approx(0.1, rel=1e-7)
    assert f.getNumParticles() == 0
    assert (P.AGBNPForce.CutoffNonPeriodic, P.AGBNPForce.CutoffPeriodic) == (1, 2)


def test_particle_accessors_and_errors():
    f = P.AGBNPForce()
    assert f.addParticle(0.17, 48.9, -1.2, 0.3, False) == 0
    assert f.addParticle(0.12, 0.0, 0.0, 0.1, True) == 1
    assert f.getNumParticles() == 2
    assert f.getParticleParameters(0) == (0.17, 48.9, -1.2, 0.3, False)
    f.setParticleParameters(1, 0.125, 0.0, 0.0, -0.1, True)
    assert f.getParticleParameters(1) == (0.125, 0.0, 0.0, -0.1, True)
    with pytest.raises(P.OpenMMException):
        f.getParticleParameters(2)
    with pytest.raises(P.OpenMMException):
        f.setParticleParameters(-1, 0.1, 0, 0, 0, False)
    for v in (0, 1, 2):
        f.setVersion(v)
        assert f.getVersion() == v
    for bad in (-1, 3):
        with pytest.raises(P.OpenMMException, match="illegal version number"):
            f.setVersion(bad)
    f.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    f.setCutoffDistance(1.2)
    assert f.getNonbondedMethod() == 1 and f.getCutoffDistance() == 1.2


def test_create_reports_parameter_errors_before_touching_the_device(systems):
    s = systems("fixture264")
    g = s.gamma.copy()
    g[np.flatnonzero(s.ishydrogen == 0)[5]] += 1.0
    force = P.AGBNPForce.from_arrays(s.radius, g, s.alpha, s.charge, s.ishydrogen)
    with pytest.raises(P.OpenMMException, match="does not support multiple gamma values"):
        P.AGBNPContext(force)
    force2 = P.AGBNPForce.from_arrays(*s.params())
    force2.setVersion(2)
    with pytest.raises(P.OpenMMException, match="version 2"):
        P.AGBNPContext(force2)


def test_no_cpu_fallback_without_device(systems):
    if _lib.load().agbnp_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    s = systems("fixture264")
    with pytest.raises(P.OpenMMException, match="no HIP device"):
        P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params()))


@pytest.mark.parametrize("name", ["fixture264", "trpcage", "1dwc"])
def test_host_i4_tables_match_oracle(systems, name):
    from oracle import Oracle
    s = systems(name)
    t = P.host_tables(s.radius, s.ishydrogen)
    o = Oracle(*s.params(), version=1).tables()
    assert t["y"].shape == o["y"].shape
    np.testing.assert_array_equal(t["type_screened"], o["type_screened"])
    np.testing.assert_array_equal(t["type_screener"], o["type_screener"])
    np.testing.assert_allclose(t["y"], o["y"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(t["y2"], o["y2"], rtol=1e-11, atol=1e-13)
    # tables vanish at the 2.0 nm end (C2 switch, AGBNPUtils.cpp:102-130)
    assert np.abs(t["y"][:, :, -1]).max() < 1e-12


def test_radius_typing_truncates_to_1e4(systems):
    r = np.array([0.17, 0.170004, 0.1701, 0.12, 0.12])
    h = np.array([0, 0, 0, 1, 1], dtype=np.int32)
    t = P.host_tables(r, h)
    # long(r*10000): 0.17 and 0.170004 share a class; 0.1701 is its own; hydrogens never screen
    assert t["type_screened"][0] == t["type_screened"][1] != t["type_screened"][2]
    assert list(t["type_screener"][3:]) == [-1, -1]
    assert t["y"].shape[:2] == (3, 2)


def test_structure_parser_units(systems):
    s = systems("fixture264")
    # TestReferenceAGBNPForce.cpp:57-69: A -> nm, kcal/mol/A^2 -> kJ/mol/nm^2
    assert s.n == 264 and s.nheavy == 136
    assert s.pos[0, 0] == pytest.approx(-0.7364)
    assert s.radius[1] == pytest.approx(0.165)
    assert s.gamma[1] == pytest.approx(0.117 * 418.4)
    assert s.alpha[1] < 0
    j = s.jittered(3)
    assert j.shape == s.pos.shape and 0 < np.abs(j - s.pos).max() < 0.02
    np.testing.assert_array_equal(j, s.jittered(3))


@pytest.mark.parametrize("name,dms", [("trpcage", "trpcage_agbnp1.dms"), ("1dwc", "1dwc_agbnp1.dms")])
def test_dms_loader_matches_exported_fixture(systems, name, dms):
    """Build-container only: the reference's example/*.dms files exist only there (never on the GPU box)."""
    path = os.path.join("/root/reference/example", dms)
    if not os.path.exists(path):
        pytest.skip("reference examples are not present on this machine")
    a, b = P.load_dms(path), systems(name)
    assert a.n == b.n
    for x, y in ((a.pos, b.pos), (a.radius, b.radius), (a.gamma, b.gamma), (a.alpha, b.alpha), (a.charge, b.charge)):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a.ishydrogen, b.ishydrogen)


def test_cxx_host_mirror_compiles_and_links():
    """cpp/AGBNPForce.h + the C++ test program build against the shared library (no GPU needed to link)."""
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "openmm_agbnp_plugin_amd")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "TestHipAGBNPForce")
        subprocess.run(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cxx", "TestHipAGBNPForce.cpp"), "-o", exe,
                        os.path.join(libdir, "libagbnp_hip.so"), f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
        if _lib.load().agbnp_hip_device_count() == 0:
            # without a device the program must fail loudly with the engine's message, not fall back to anything
            out = subprocess.run([exe, "1"], input="1\n0 0 0 0 1.7 0.1 0.117 0\n", text=True, capture_output=True, timeout=60)
            assert out.returncode != 0 and "no HIP device" in out.stdout


def test_top_level_module_keeps_the_reference_import_line():
    """The reference's scripts say `from AGBNPplugin import AGBNPForce` (python/AGBNPPlugin.i:1: %module AGBNPplugin)."""
    import AGBNPplugin
    assert AGBNPplugin.AGBNPForce is P.AGBNPForce
    for method in ("getNumParticles", "addParticle", "setParticleParameters", "updateParametersInContext", "setCutoffDistance",
                   "setNonbondedMethod", "setVersion", "getParticleParameters"):  # the SWIG interface's method list (:46-84)
        assert callable(getattr(AGBNPplugin.AGBNPForce, method))
    assert (AGBNPplugin.AGBNPForce.NoCutoff, AGBNPplugin.AGBNPForce.CutoffNonPeriodic, AGBNPplugin.AGBNPForce.CutoffPeriodic) == (0, 1, 2)


def test_unit_quantities_are_stripped_like_the_swig_layer():
    """OpenMM's SWIG typemaps hand the C++ class the value in the MD unit system; anything with
    value_in_unit_system() is treated that way (no units module is needed for plain numbers)."""
    class Quantity:  # stand-in for openmm.unit.Quantity
        def __init__(self, value, to_md):
            self.value, self.to_md = value, to_md

        def value_in_unit_system(self, system):
            return self.value * self.to_md

    f = P.AGBNPForce()
    f.addParticle(Quantity(1.7, 0.1), Quantity(0.117, 418.4), -1.2, Quantity(0.3, 1.0), False)  # Angstrom, kcal/mol/A^2
    r, g, a, q, h = f.getParticleParameters(0)
    assert r == pytest.approx(0.17) and g == pytest.approx(48.9528) and (a, q, h) == (-1.2, 0.3, False)
    f.setParticleParameters(0, Quantity(2.0, 0.1), 1.0, 2.0, 3.0, True)
    assert f.getParticleParameters(0) == (pytest.approx(0.2), 1.0, 2.0, 3.0, True)
    f.setCutoffDistance(Quantity(12.0, 0.1))
    assert f.getCutoffDistance() == pytest.approx(1.2)


def test_bench_instruction_table_tracks_the_counters():
    """bench.py prices the pair kernels' vector-issue roof with SQ_INSTS_VALU of the committed counter pass; its hand-read
    fallback table (instructions per wave-step x wave-steps of the geometry) must stay within 3 % of those counters, or it
    has rotted with a kernel edit."""
    import bench
    import openmm_agbnp_plugin_amd as P
    counters, source = bench.counter_valu_instructions("1dwc")
    if counters is None:
        pytest.skip("no counter summary under profiles/")
    s = P.load_system("1dwc")
    steps = bench.pair_wave_steps(s, s.jittered(1020))
    for kernel, kinds in bench.PAIR_STEP_VALU.items():
        if kernel not in counters:
            continue  # (a kernel that the profiled configuration does not launch)
        model = sum(steps[kernel][kind] * valu for kind, (valu, _) in kinds.items())
        assert abs(model / counters[kernel]["valu"] - 1.0) < 0.03, (kernel, model, counters[kernel]["valu"], source)


def test_profiles_say_what_they_were_measured_on():
    """the newest committed profile carries its own head (git + the library's build id = SHA-256 of the
    engine's sources), bench.py repeats it in the line, and the summaries' kernel names are the engine's (a template argument
    more or less does not split a kernel's rows)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    d = bench.newest_profile_dir()
    assert d is not None and os.path.exists(os.path.join(d, "profile_head.json")), d
    head = json.load(open(os.path.join(d, "profile_head.json")))
    assert len(head["library_build_id"]) == 16 and head["git_head"]
    rec = bench.profile_head()
    assert rec["library_build_id"] == head["library_build_id"] and rec["profile"].startswith("profiles/r")
    traffic = json.load(open(os.path.join(root, "profiles", "traffic_pmc.json")))
    assert traffic["profile_head"]["library_build_id"] == head["library_build_id"] and traffic["kernel"] == "k_tree_cavity"
    import csv
    kernels = {r["kernel"] for r in csv.DictReader(open(os.path.join(d, "pmc_utilization.csv")))}
    assert {"k_tree_cavity", "k_gb_tiles", "k_rows<0>", "k_rows<1>", "k_tree_pseudo"} <= kernels, kernels
    counters, source = bench.counter_valu_instructions("1dwc")
    assert counters and {"k_gb_tiles", "k_born_rows", "k_dborn_rows"} <= set(counters) and os.path.basename(d) in source
