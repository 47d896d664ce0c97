"""ctypes binding of libagbnp_hip.so (C ABI: include/agbnp_hip.h).  There is no fallback: if the shared
library is missing the import of any compute entry point raises, loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagbnp_hip.so")

OK, ERR_INVALID_ARGUMENT, ERR_PARAMETERS, ERR_DEVICE, ERR_CAPACITY = 0, 1, 2, 3, 4

# every symbol include/agbnp_hip.h declares
SYMBOLS = [
    "agbnp_hip_create", "agbnp_hip_update_parameters", "agbnp_hip_execute_host", "agbnp_hip_execute_device",
    "agbnp_hip_finish", "agbnp_hip_get_scalar", "agbnp_hip_get_vector", "agbnp_hip_get_table_sizes",
    "agbnp_hip_get_tables", "agbnp_hip_host_tables", "agbnp_hip_num_particles", "agbnp_hip_version",
    "agbnp_hip_last_error", "agbnp_hip_destroy", "agbnp_hip_device_count",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
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
    lib.agbnp_hip_finish.argtypes = [vp, vp, ip]
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
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the library does not export it
    _lib = lib
    return lib


def last_error(handle=None):
    msg = load().agbnp_hip_last_error(handle)
    return msg.decode() if msg else ""
