"""The OpenMM platform plugin of the engine (openmm_glue/HipAGBNPKernels.cpp): compiled against a test double of the
OpenMM API (tests/openmm_mock; OpenMM is not in the image), against the reference's own API headers where
/root/reference exists, and -- on the GPU box -- run end to end: System + AGBNPForce -> Context -> ForceImpl ->
Platform("HIP").createKernel("CalcAGBNPForce") -> initialize -> execute, with the context's atoms shuffled and padded,
forces read back from the 2^32 fixed-point buffer, against the reference's printed known answers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "openmm_mock")
GLUE = os.path.join(ROOT, "openmm_glue", "HipAGBNPKernels.cpp")
LIBDIR = os.path.join(ROOT, "openmm_agbnp_plugin_amd")
REF_API = "/root/reference/openmmapi/include"
COMMON = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{MOCK}", f"-I{ROOT}/include", f"-I{ROOT}/openmm_glue"]


def build_test_program(tmp_path):
    exe = str(tmp_path / "TestHipPlatformAGBNPForce")
    subprocess.run(COMMON + ["-O1", f"-I{MOCK}/agbnp_api", os.path.join(ROOT, "tests", "cxx", "TestHipPlatformAGBNPForce.cpp"), GLUE,
                             os.path.join(LIBDIR, "libagbnp_hip.so"), "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}",
                             "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_glue_compiles_against_the_openmm_test_double():
    subprocess.run(COMMON + ["-fsyntax-only", "-Wall", f"-I{MOCK}/agbnp_api", GLUE], check=True)


@pytest.mark.skipif(not os.path.isdir(REF_API), reason="the reference tree only exists in the build container")
def test_glue_compiles_against_the_reference_api_headers_in_place():
    """Same source, but AGBNPForce.h / AGBNPKernels.h are the reference's own files (nothing is copied)."""
    subprocess.run(COMMON + ["-fsyntax-only", f"-I{REF_API}", GLUE], check=True)


def test_glue_exports_the_plugin_entry_points(tmp_path):
    """What OpenMM's plugin loader looks up in a platform plugin (ReferenceAGBNPKernelFactory.cpp:14-36)."""
    so = str(tmp_path / "libAGBNPPluginHip.so")
    subprocess.run(COMMON + ["-O1", "-shared", "-fPIC", f"-I{MOCK}/agbnp_api", GLUE, os.path.join(LIBDIR, "libagbnp_hip.so"),
                             "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", so], check=True)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    for name in ("registerPlatforms", "registerKernelFactories", "registerAGBNPHipKernelFactories"):
        assert f" T {name}" in syms


def test_plugin_path_fails_cleanly_without_a_device(tmp_path):
    """No GPU: the test program must end with an OpenMM-style exception, not a crash (CPU-only check of the wiring)."""
    from openmm_agbnp_plugin_amd import _lib
    if _lib.load().agbnp_hip_device_count() > 0:
        pytest.skip("a HIP device is present")
    exe = build_test_program(tmp_path)
    data = open(os.path.join(ROOT, "openmm_agbnp_plugin_amd", "data", "fixture264.dat")).read()
    out = subprocess.run([exe, "1", "double"], input=data, text=True, capture_output=True, timeout=120)
    assert out.returncode == 2 and out.stdout.startswith("exception:")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["double", "mixed", "single"])
@pytest.mark.parametrize("version", [0, 1])
def test_plugin_path_reproduces_the_reference_known_answers(gpu_required, tmp_path, version, precision):
    from tests.pins import REFERENCE_PRINTED
    exe = build_test_program(tmp_path)
    data = open(os.path.join(ROOT, "openmm_agbnp_plugin_amd", "data", "fixture264.dat")).read()
    out = subprocess.run([exe, str(version), precision], input=data, text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.split("\n")
    want = REFERENCE_PRINTED[version]
    if precision == "single":  # float positions and a float energy accumulator: 7 digits of a ~2500 kJ/mol energy
        assert abs(float(lines[0].split()[1]) - want["energy"]) < 0.05
    else:
        assert lines[0] == f"Energy: {want['energy']:g}"
        assert lines[1] == f"Energy: {want['energy_moved']:g}"
        assert f"Energy Change from Gradient: {want['change_from_gradient']:g}" in out.stdout
    assert "PASS" in out.stdout


@pytest.mark.gpu
def test_plugin_path_with_the_reference_blocking_check(gpu_required, tmp_path):
    """AGBNP_HIP_CHECK_MODE=finish: the reference's own protocol (stream synchronisation and a read of the overflow log after
    every evaluation) instead of the default wait for the device's verdict word; same numbers."""
    from tests.pins import REFERENCE_PRINTED
    exe = build_test_program(tmp_path)
    data = open(os.path.join(ROOT, "openmm_agbnp_plugin_amd", "data", "fixture264.dat")).read()
    env = dict(os.environ, AGBNP_HIP_CHECK_MODE="finish")
    out = subprocess.run([exe, "1", "double"], input=data, text=True, capture_output=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split("\n")[0] == f"Energy: {REFERENCE_PRINTED[1]['energy']:g}"
    assert "PASS" in out.stdout


@pytest.mark.gpu
def test_plugin_path_in_poll_mode(gpu_required, tmp_path):
    """AGBNP_HIP_CHECK_MODE=poll: execute() looks at the engine's pinned status words instead of synchronising the stream
    every step; the numbers of a healthy run are the strict mode's."""
    from tests.pins import REFERENCE_PRINTED
    exe = build_test_program(tmp_path)
    data = open(os.path.join(ROOT, "openmm_agbnp_plugin_amd", "data", "fixture264.dat")).read()
    env = dict(os.environ, AGBNP_HIP_CHECK_MODE="poll")
    out = subprocess.run([exe, "1", "double"], input=data, text=True, capture_output=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.split("\n")[0] == f"Energy: {REFERENCE_PRINTED[1]['energy']:g}"
    assert "PASS" in out.stdout
