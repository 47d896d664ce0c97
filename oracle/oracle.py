"""ctypes loader for oracle/_build/libagbnp_oracle.so (TEST INFRASTRUCTURE, see agbnp_oracle.cpp header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def oracle_lib_path():
    return os.path.join(_HERE, "_build", "libagbnp_oracle.so")


def build_oracle(force=False):
    """Compile the oracle with the committed Makefile (g++ only; no reference sources involved)."""
    so = oracle_lib_path()
    src = os.path.join(_HERE, "agbnp_oracle.cpp")
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


_lib = None


def _load():
    global _lib
    if _lib is None:
        so = build_oracle()
        lib = C.CDLL(so)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_long)
        lib.agbnp_oracle_create.restype = C.c_void_p
        lib.agbnp_oracle_create.argtypes = [C.c_int, dp, dp, dp, dp, ip, C.c_int, C.c_char_p, C.c_int]
        lib.agbnp_oracle_destroy.argtypes = [C.c_void_p]
        lib.agbnp_oracle_set_cutoff.argtypes = [C.c_void_p, C.c_double]
        lib.agbnp_oracle_update.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, ip, C.c_char_p, C.c_int]
        lib.agbnp_oracle_execute.argtypes = [C.c_void_p, dp, dp, dp]
        lib.agbnp_oracle_scalar.restype = C.c_double
        lib.agbnp_oracle_scalar.argtypes = [C.c_void_p, C.c_int]
        lib.agbnp_oracle_vector.argtypes = [C.c_void_p, C.c_int, dp]
        lib.agbnp_oracle_tree_stats.argtypes = [C.c_void_p, lp, lp, lp]
        lib.agbnp_oracle_tables.argtypes = [C.c_void_p, dp, dp, dp, ip, ip]
        _lib = lib
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class OracleError(RuntimeError):
    pass


class Oracle:
    """One 'kernel' of the CPU restatement: initialize once (tables), execute per geometry."""

    SCALARS = dict(e_vol1=0, e_vol2=1, e_gb=2, e_vdw=3, volume1=4, volume2=5, slots=6, nscreened=7, nscreener=8)
    VECTORS = dict(selfvol_large=0, selfvol_vdw=1, born=2, scale=3, brw=4, bru=5, Y=6, W=7, U=8, freevol_vdw=9)

    def __init__(self, radius, gamma, alpha, charge, ishydrogen, version=1, cutoff=None, method=1):
        """cutoff=None: the Reference platform's semantics (the pinned oracle).  cutoff=c: the FAST-mode restatement of
        the reference's OpenCL platform -- every AGBNP1 pair loop truncated at r < c (parity unpinned, see the .cpp).
        method: the force's nonbonded method (0 NoCutoff, 1 CutoffNonPeriodic, 2 CutoffPeriodic); as on that platform the
        cutoff only exists for a method other than NoCutoff (OpenCLAGBNPKernels.cpp:487), and CutoffPeriodic is not restated."""
        if cutoff is not None and method == 2:
            raise OracleError("the fast-mode switch does not restate CutoffPeriodic")
        if method == 0:
            cutoff = None
        lib = _load()
        self.n = len(radius)
        self._p = [_d(radius), _d(gamma), _d(alpha), _d(charge), np.ascontiguousarray(ishydrogen, dtype=np.int32)]
        err = C.create_string_buffer(512)
        self._h = lib.agbnp_oracle_create(self.n, _dp(self._p[0]), _dp(self._p[1]), _dp(self._p[2]), _dp(self._p[3]),
                                          _ip(self._p[4]), int(version), err, 512)
        if not self._h:
            raise OracleError(err.value.decode())
        self.version = version
        if cutoff is not None:
            lib.agbnp_oracle_set_cutoff(self._h, float(cutoff))

    def update(self, radius, gamma, alpha, charge, ishydrogen):
        p = [_d(radius), _d(gamma), _d(alpha), _d(charge), np.ascontiguousarray(ishydrogen, dtype=np.int32)]
        err = C.create_string_buffer(512)
        rc = _load().agbnp_oracle_update(self._h, len(p[0]), _dp(p[0]), _dp(p[1]), _dp(p[2]), _dp(p[3]), _ip(p[4]), err, 512)
        if rc != 0:
            raise OracleError(err.value.decode())

    def execute(self, pos, force_accum=None):
        """Returns (energy, forces).  forces = force_accum + F (the Reference platform accumulates)."""
        pos = _d(pos).reshape(self.n, 3)
        f = np.zeros((self.n, 3)) if force_accum is None else _d(force_accum).reshape(self.n, 3).copy()
        e = C.c_double(0)
        _load().agbnp_oracle_execute(self._h, _dp(pos), _dp(f), C.byref(e))
        return e.value, f

    def scalar(self, name):
        return _load().agbnp_oracle_scalar(self._h, self.SCALARS[name])

    def vector(self, name):
        out = np.zeros(self.n)
        rc = _load().agbnp_oracle_vector(self._h, self.VECTORS[name], _dp(out))
        if rc != 0:
            raise OracleError(f"vector {name} not available (run execute first / version 1 only)")
        return out

    def tree_stats(self):
        counts = np.zeros(9, dtype=np.int64)
        ms, mc = C.c_long(0), C.c_long(0)
        _load().agbnp_oracle_tree_stats(self._h, counts.ctypes.data_as(C.POINTER(C.c_long)), C.byref(ms), C.byref(mc))
        return dict(level_counts=counts.tolist(), max_subtree=ms.value, max_children=mc.value)

    def tables(self):
        ni, nj = int(self.scalar("nscreened")), int(self.scalar("nscreener"))
        x = np.zeros(16)
        y = np.zeros((ni * nj, 16))
        y2 = np.zeros((ni * nj, 16))
        ti = np.zeros(self.n, dtype=np.int32)
        tj = np.zeros(self.n, dtype=np.int32)
        _load().agbnp_oracle_tables(self._h, _dp(x), _dp(y), _dp(y2), _ip(ti), _ip(tj))
        return dict(x=x, y=y.reshape(ni, nj, 16), y2=y2.reshape(ni, nj, 16), type_screened=ti, type_screener=tj)

    def __del__(self):
        try:
            if self._h:
                _load().agbnp_oracle_destroy(self._h)
                self._h = None
        except Exception:
            pass
