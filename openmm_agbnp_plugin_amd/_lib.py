"""ctypes binding of libagbnp_hip.so (C ABI: include/agbnp_hip.h).  There is no fallback: if the shared
library is missing the import of any compute entry point raises, loudly."""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGBNP_HIP_LIBRARY") or os.path.join(_HERE, "libagbnp_hip.so")  # override: diagnostic builds only

OK, ERR_INVALID_ARGUMENT, ERR_PARAMETERS, ERR_DEVICE, ERR_CAPACITY, ERR_TIMEOUT = 0, 1, 2, 3, 4, 5

# every symbol include/agbnp_hip.h declares
SYMBOLS = [
    "agbnp_hip_create", "agbnp_hip_update_parameters", "agbnp_hip_execute_host", "agbnp_hip_execute_device",
    "agbnp_hip_execute_openmm", "agbnp_hip_atom_order_changed", "agbnp_hip_finish", "agbnp_hip_poll", "agbnp_hip_wait_verdict", "agbnp_hip_withheld_evaluations", "agbnp_hip_generation", "agbnp_hip_get_scalar", "agbnp_hip_get_vector", "agbnp_hip_get_table_sizes",
    "agbnp_hip_get_tables", "agbnp_hip_host_tables", "agbnp_hip_num_particles", "agbnp_hip_version",
    "agbnp_hip_last_error", "agbnp_hip_destroy", "agbnp_hip_device_count", "agbnp_hip_build_id",
    "agbnp_hip_set_mode", "agbnp_hip_get_mode", "agbnp_hip_set_diagnostics", "agbnp_hip_set_profiling", "agbnp_hip_num_kernels", "agbnp_hip_kernel_name", "agbnp_hip_get_kernel_times",
]

_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7) and asks for it by file name, so if this library pulled in /opt/rocm's copy
    first, a later `import torch` would load a second runtime and find no GPU.  Loading torch's copy first
    (without importing torch) makes both resolve to the same object.  Set AGBNP_HIP_SYSTEM_RUNTIME=1 to
    keep the system runtime (processes that never import torch)."""
    if os.environ.get("AGBNP_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  openmm_agbnp_plugin_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    lib.agbnp_hip_create.argtypes = [C.POINTER(vp), C.c_int, dp, dp, dp, dp, ip, C.c_int, C.c_int, C.c_double, C.c_int]
    lib.agbnp_hip_update_parameters.argtypes = [vp, C.c_int, dp, dp, dp, dp, ip]
    lib.agbnp_hip_execute_host.argtypes = [vp, dp, dp, dp]
    lib.agbnp_hip_execute_device.argtypes = [vp, vp, vp, vp, vp]
    lib.agbnp_hip_execute_openmm.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]
    lib.agbnp_hip_atom_order_changed.argtypes = [vp]
    lib.agbnp_hip_finish.argtypes = [vp, vp, ip]
    lib.agbnp_hip_poll.argtypes = [vp, ip, ip]
    lib.agbnp_hip_wait_verdict.argtypes = [vp, C.c_int, C.c_double, ip, ip]
    lib.agbnp_hip_withheld_evaluations.argtypes = [vp, ip, C.c_int]
    lib.agbnp_hip_generation.argtypes = [vp]
    lib.agbnp_hip_generation.restype = C.c_uint
    lib.agbnp_hip_get_scalar.argtypes = [vp, C.c_int, dp]
    lib.agbnp_hip_get_vector.argtypes = [vp, C.c_int, dp]
    lib.agbnp_hip_get_table_sizes.argtypes = [vp, ip, ip]
    lib.agbnp_hip_get_tables.argtypes = [vp, dp, dp, ip, ip]
    lib.agbnp_hip_host_tables.argtypes = [C.c_int, dp, ip, ip, ip, dp, dp, C.c_int, ip, ip]
    lib.agbnp_hip_num_particles.argtypes = [vp]
    lib.agbnp_hip_version.argtypes = [vp]
    lib.agbnp_hip_last_error.argtypes = [vp]
    lib.agbnp_hip_last_error.restype = C.c_char_p
    lib.agbnp_hip_destroy.argtypes = [vp]
    lib.agbnp_hip_destroy.restype = None
    lib.agbnp_hip_device_count.argtypes = []
    lib.agbnp_hip_build_id.argtypes = []
    lib.agbnp_hip_build_id.restype = C.c_char_p
    lib.agbnp_hip_set_mode.argtypes = [vp, C.c_int]
    lib.agbnp_hip_get_mode.argtypes = [vp]
    lib.agbnp_hip_set_diagnostics.argtypes = [vp, C.c_int]
    lib.agbnp_hip_set_profiling.argtypes = [vp, C.c_int]
    lib.agbnp_hip_num_kernels.argtypes = []
    lib.agbnp_hip_kernel_name.argtypes = [C.c_int]
    lib.agbnp_hip_kernel_name.restype = C.c_char_p
    lib.agbnp_hip_get_kernel_times.argtypes = [vp, dp, C.POINTER(C.c_long)]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the library does not export it
    _lib = lib
    return lib


def build_id():
    """First 16 hex digits of the SHA-256 of the sources the loaded library was built from (csrc/Makefile)."""
    return load().agbnp_hip_build_id().decode()


def last_error(handle=None):
    msg = load().agbnp_hip_last_error(handle)
    return msg.decode() if msg else ""
