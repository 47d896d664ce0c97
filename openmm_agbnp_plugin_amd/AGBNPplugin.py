"""Host-side mirror of the reference's Python module `AGBNPplugin` (python/AGBNPPlugin.i:46-84) and of the
plugin's kernel interface, on top of the gfx950 engine's C ABI.

`AGBNPForce` keeps the reference's method names, argument order and meaning, defaults and error
behaviour (openmmapi/include/AGBNPForce.h:39-155, openmmapi/src/AGBNPForce.cpp:15-78).
`HipCalcAGBNPForceKernel` plays the role of a platform's `CalcAGBNPForceKernel`
(openmmapi/include/AGBNPKernels.h:19-47): initialize / execute / copyParametersToContext.
`AGBNPContext` is the small stand-in for the OpenMM Context + ForceImpl pair that the parity tests
and benchmarks need (OpenMM itself is not part of this path): it owns positions and calls the kernel
the way AGBNPForceImpl does (openmmapi/src/AGBNPForceImpl.cpp:27-46).
"""
import ctypes as C

import numpy as np

from . import _lib


class OpenMMException(Exception):
    """Same role as OpenMM::OpenMMException in the reference: every API error surfaces as this type."""


def _md_unit_system():
    """OpenMM's md_unit_system (nm, ps, amu, kJ/mol, e, K) if a units module is importable in this process."""
    for name in ("openmm.unit", "simtk.unit"):
        try:
            module = __import__(name, fromlist=["md_unit_system"])
            return module.md_unit_system
        except Exception:
            continue
    return "md_unit_system"


def _strip_units(value):
    """The SWIG layer of the reference accepts OpenMM unit Quantities for every double argument and hands the C++
    class the bare number in the MD unit system (python/AGBNPPlugin.i:3-4: OpenMM's swig/typemaps.i calls
    value_in_unit_system(md_unit_system)).  Duck-typed here: anything with that method is converted, plain numbers pass."""
    if hasattr(value, "value_in_unit_system"):
        return float(value.value_in_unit_system(_md_unit_system()))
    return float(value)


class AGBNPForce:
    # enum NonbondedMethod (openmmapi/include/AGBNPForce.h:44-59)
    NoCutoff = 0
    CutoffNonPeriodic = 1
    CutoffPeriodic = 2

    def __init__(self):
        # defaults of AGBNPForce::AGBNPForce() (openmmapi/src/AGBNPForce.cpp:15)
        self._particles = []
        self._method = AGBNPForce.NoCutoff
        self._cutoff = 1.0
        self._version = 1
        self._solvent_radius = 1.0 * float(np.float32(0.1))  # SOLVENT_RADIUS (1.0*ANG)
        self._force_group = 0

    def getNumParticles(self):
        return len(self._particles)

    def addParticle(self, radius, gamma, vdw_alpha, charge, ishydrogen):
        """radius nm, gamma kJ/mol/nm^2, vdw_alpha kJ/mol nm^3... (as the reference), charge e. Returns the index."""
        self._particles.append([_strip_units(radius), _strip_units(gamma), _strip_units(vdw_alpha), _strip_units(charge), bool(ishydrogen)])
        self._cache = None
        return len(self._particles) - 1

    def _check_index(self, index):
        if index < 0 or index >= len(self._particles):
            raise OpenMMException("Assertion failure: Index out of range")  # ASSERT_VALID_INDEX

    def setParticleParameters(self, index, radius, gamma, vdw_alpha, charge, ishydrogen):
        self._check_index(index)
        row = [_strip_units(radius), _strip_units(gamma), _strip_units(vdw_alpha), _strip_units(charge), bool(ishydrogen)]
        self._particles[index] = row
        cache = getattr(self, "_cache", None)
        if cache is not None:  # the arrays that cross the boundary follow in place (updateParametersInContext after a sweep of these)
            for k in range(4):
                cache[k][index] = row[k]
            cache[4][index] = 1 if row[4] else 0

    def getParticleParameters(self, index):
        """Returns the tuple (radius, gamma, vdw_alpha, charge, ishydrogen), as the SWIG wrapper does."""
        self._check_index(index)
        return tuple(self._particles[index])

    def getNonbondedMethod(self):
        return self._method

    def setNonbondedMethod(self, method):
        self._method = int(method)

    def getCutoffDistance(self):
        return self._cutoff

    def setCutoffDistance(self, distance):
        self._cutoff = _strip_units(distance)

    def getSolventRadius(self):
        return self._solvent_radius

    def setVersion(self, agbnp_version):
        if 0 <= agbnp_version <= 2:
            self._version = int(agbnp_version)
        else:
            raise OpenMMException("AGBNPForce::setVersion(): illegal version number")

    def getVersion(self):
        return self._version

    def getForceGroup(self):
        return self._force_group

    def setForceGroup(self, group):
        self._force_group = int(group)

    def updateParametersInContext(self, context):
        context._update_parameters(self)

    # helpers for the engine boundary
    def _arrays(self):
        """The five parameter arrays as they cross the C ABI (built once, then kept up to date by setParticleParameters)."""
        cache = getattr(self, "_cache", None)
        if cache is None:
            p = self._particles
            f = lambda k: np.ascontiguousarray([x[k] for x in p], dtype=np.float64)
            cache = self._cache = (f(0), f(1), f(2), f(3), np.ascontiguousarray([1 if x[4] else 0 for x in p], dtype=np.int32))
        return cache

    @classmethod
    def from_arrays(cls, radius, gamma, vdw_alpha, charge, ishydrogen, version=1):
        force = cls()
        force.setVersion(version)
        for r, g, a, q, h in zip(radius, gamma, vdw_alpha, charge, ishydrogen):
            force.addParticle(r, g, a, q, bool(h))
        return force


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class HipCalcAGBNPForceKernel:
    """The 'HIP platform' implementation of CalcAGBNPForceKernel."""

    @staticmethod
    def Name():
        return "CalcAGBNPForce"

    MODES = {"reference": 0, "fast": 1, "deterministic": 2, "fast+deterministic": 3, "fast+single": 5}

    def __init__(self, device=0, mode="reference"):
        """mode "reference" (default): the Reference platform's semantics, the parity target.  mode "fast": the
        OpenCL platform's semantics -- every pair stage truncated at the force's cutoff distance (include/agbnp_hip.h)."""
        self._h = None
        self._device = device
        self._mode = self.MODES[mode]
        self.numParticles = 0

    def initialize(self, force):
        lib = _lib.load()
        self.release()
        r, g, a, q, h = force._arrays()
        self.numParticles = len(r)
        handle = C.c_void_p()
        rc = lib.agbnp_hip_create(C.byref(handle), len(r), _dp(r), _dp(g), _dp(a), _dp(q), _ip(h), force.getVersion(),
                                  force.getNonbondedMethod(), force.getCutoffDistance(), self._device)
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(None))
        self._h = handle
        if self._mode and lib.agbnp_hip_set_mode(self._h, self._mode) != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))

    def set_mode(self, mode):
        self._need()
        self._mode = self.MODES[mode]
        if _lib.load().agbnp_hip_set_mode(self._h, self._mode) != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))

    def _need(self):
        if self._h is None:
            raise OpenMMException("HipCalcAGBNPForceKernel: initialize() has not been called")

    def execute(self, positions, forces, includeForces=True, includeEnergy=True):
        """CPU-platform convention of the reference: forces (N x 3 float64 array) are accumulated in place,
        the energy is returned.  includeForces/includeEnergy are accepted and ignored, as in the reference
        (ReferenceAGBNPKernels.cpp:139-149 always computes both)."""
        self._need()
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1)
        if pos.size != 3 * self.numParticles:
            raise OpenMMException("execute(): wrong number of positions")
        if not (isinstance(forces, np.ndarray) and forces.dtype == np.float64 and forces.flags.c_contiguous
                and forces.size == 3 * self.numParticles):
            raise OpenMMException("execute(): forces must be a C-contiguous float64 array of 3N values")
        e = C.c_double(0.0)
        rc = _lib.load().agbnp_hip_execute_host(self._h, _dp(pos), _dp(forces), C.byref(e))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))
        return e.value

    def execute_device(self, d_positions, d_forces, d_energy, stream=None):
        """GPU-platform convention (reference OpenCL platform): raw FP64 device pointers (ints), forces and
        energy are ADDED on the device, nothing is returned.  Asynchronous; call finish().

        What a caller has to know (include/agbnp_hip.h, "Five-launch mode"): the tree launch finds every heavy atom's neighbours
        through masks laid down at an earlier evaluation, with a skin.  Positions that differ from the PREVIOUS evaluation's by
        more than 0.04 nm for some heavy atom (a minimiser's long step, a Monte-Carlo move, unrelated geometries one after the
        other) cost one withheld evaluation: finish() reports it, withheld() names it, running it again is right (execute(), the
        host-buffer entry point, repeats by itself).  MD steps are two orders of magnitude below that;
        AGBNP_HIP_FIVE_LAUNCHES=0 lifts the restriction for the price of one more launch per evaluation."""
        self._need()
        rc = _lib.load().agbnp_hip_execute_device(self._h, C.c_void_p(d_positions), C.c_void_p(d_forces), C.c_void_p(d_energy),
                                                  C.c_void_p(stream or 0))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))

    def execute_openmm(self, d_posq, posq_is_double, d_posq_correction, d_atom_index, padded_num_atoms, d_force_buffer,
                       d_energy_buffer, energy_is_double, energy_slot=0, stream=None):
        """The data conventions of an OpenMM GPU context (reference OpenCL platform): posq real4 in the context's
        atom order (+ optional float4 correction), atomIndex map, 2^32 fixed-point force planes, energy buffer slot.
        Raw device pointers (ints; 0 = NULL).  Asynchronous; call finish()."""
        self._need()
        rc = _lib.load().agbnp_hip_execute_openmm(self._h, C.c_void_p(d_posq), int(bool(posq_is_double)), C.c_void_p(d_posq_correction or 0),
                                                  C.c_void_p(d_atom_index or 0), int(padded_num_atoms), C.c_void_p(d_force_buffer),
                                                  C.c_void_p(d_energy_buffer or 0), int(bool(energy_is_double)), int(energy_slot),
                                                  C.c_void_p(stream or 0))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))

    def atom_order_changed(self):
        """The context has reordered its atoms (same atomIndex array, new contents): the next execute_openmm() rebuilds the
        engine's maps first instead of losing one evaluation to the check on the device."""
        self._need()
        _lib.load().agbnp_hip_atom_order_changed(self._h)

    def finish(self, stream=None):
        """Synchronise and read the device's overflow log: returns the number of evaluations enqueued since the
        previous finish() whose forces and energy were WITHHELD on the device (0 = all complete).  Those must be
        run again (see withheld()); the context has already switched to the packing / capacity variant they need."""
        self._need()
        rep = C.c_int(0)
        rc = _lib.load().agbnp_hip_finish(self._h, C.c_void_p(stream or 0), C.byref(rep))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))
        return int(rep.value)

    def poll(self):
        """Non-blocking: (evaluations completed since the last finish(), how many of them were withheld), read from pinned
        host memory the device writes at the end of every evaluation (no device call)."""
        self._need()
        done, bad = C.c_int(0), C.c_int(0)
        if _lib.load().agbnp_hip_poll(self._h, C.byref(done), C.byref(bad)) != _lib.OK:
            raise OpenMMException("agbnp_hip_poll: no pinned status memory")
        return int(done.value), int(bad.value)

    def wait_verdict(self, evaluations=0, timeout=10.0):
        """The strict per-evaluation check without draining the stream: blocks this thread (no device call) until the device
        has delivered its verdict on every evaluation enqueued since the last finish() (or on `evaluations` of them) and
        returns (evaluations judged, how many were withheld).  The verdict of an evaluation is final when its tree stage
        has ended; its forces follow on the stream, gated on the device by the same words.  Raises on a timeout."""
        self._need()
        done, bad = C.c_int(0), C.c_int(0)
        rc = _lib.load().agbnp_hip_wait_verdict(self._h, int(evaluations), float(timeout), C.byref(done), C.byref(bad))
        if rc == _lib.ERR_TIMEOUT:
            raise OpenMMException(f"agbnp_hip_wait_verdict: timed out after {timeout} s with {int(done.value)} evaluation(s) judged")
        if rc != _lib.OK:
            raise OpenMMException("agbnp_hip_wait_verdict: no pinned status memory")
        return int(done.value), int(bad.value)

    def withheld(self):
        """Indices (enqueue order since the finish() before the last one) of the evaluations the last finish()
        reported as withheld."""
        self._need()
        lib = _lib.load()
        n = lib.agbnp_hip_withheld_evaluations(self._h, None, 0)
        idx = np.full(max(n, 1), -1, dtype=np.int32)  # the library fills as many indices as its log holds (it only COUNTS beyond bit 2047)
        lib.agbnp_hip_withheld_evaluations(self._h, _ip(idx), n)
        return [int(k) for k in idx if k >= 0]

    def generation(self):
        """Changes when a captured HIP graph of execute_device has gone stale (capacity variant raised)."""
        self._need()
        return int(_lib.load().agbnp_hip_generation(self._h))

    def copyParametersToContext(self, force):
        self._need()
        r, g, a, q, h = force._arrays()
        rc = _lib.load().agbnp_hip_update_parameters(self._h, len(r), _dp(r), _dp(g), _dp(a), _dp(q), _ip(h))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))

    # ---- diagnostics (test support) -------------------------------------------------------------
    SCALARS = dict(e_vol1=0, e_vol2=1, e_atom=2, e_gb_pair=3, max_subtree_nodes=4, total_nodes=5, variant=6, max_local_atoms=7, forests=8, rows_on=9, row_builds=10, pack_level=11, pack_age=12, row_slice=13, pack_plans=14, overflow_kinds=15, launches=16, healed_forests=17)
    VECTORS = dict(selfvol_vdw=0, born=1, scale=2, selfvol_large=3, subtree_nodes=4, subtree_atoms=5)

    def scalar(self, name):
        self._need()
        v = C.c_double(0)
        rc = _lib.load().agbnp_hip_get_scalar(self._h, self.SCALARS[name], C.byref(v))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))
        return v.value

    def vector(self, name):
        self._need()
        out = np.zeros(self.numParticles)
        rc = _lib.load().agbnp_hip_get_vector(self._h, self.VECTORS[name], _dp(out))
        if rc != _lib.OK:
            raise OpenMMException(_lib.last_error(self._h))
        return out

    def tables(self):
        self._need()
        lib = _lib.load()
        ni, nj = C.c_int(0), C.c_int(0)
        lib.agbnp_hip_get_table_sizes(self._h, C.byref(ni), C.byref(nj))
        y = np.zeros(ni.value * nj.value * 16)
        y2 = np.zeros_like(y)
        ti = np.zeros(self.numParticles, dtype=np.int32)
        tj = np.zeros(self.numParticles, dtype=np.int32)
        lib.agbnp_hip_get_tables(self._h, _dp(y), _dp(y2), _ip(ti), _ip(tj))
        return dict(y=y.reshape(ni.value, nj.value, 16), y2=y2.reshape(ni.value, nj.value, 16), type_screened=ti, type_screener=tj)

    def set_diagnostics(self, enabled):
        """Collect the pass-1 (enlarged radii) self volumes too (vector 'selfvol_large')."""
        self._need()
        _lib.load().agbnp_hip_set_diagnostics(self._h, 1 if enabled else 0)

    def set_profiling(self, enabled):
        self._need()
        _lib.load().agbnp_hip_set_profiling(self._h, 1 if enabled else 0)

    def kernel_times(self):
        """{kernel name: (total ms, launches)} accumulated since set_profiling(True)."""
        self._need()
        lib = _lib.load()
        k = lib.agbnp_hip_num_kernels()
        ms = np.zeros(k)
        cnt = np.zeros(k, dtype=np.int64)
        lib.agbnp_hip_get_kernel_times(self._h, _dp(ms), cnt.ctypes.data_as(C.POINTER(C.c_long)))
        return {lib.agbnp_hip_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def release(self):
        if self._h is not None:
            _lib.load().agbnp_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def host_tables(radius, ishydrogen):
    """I4 tables from the engine's host code (no device needed)."""
    lib = _lib.load()
    r = np.ascontiguousarray(radius, dtype=np.float64)
    h = np.ascontiguousarray(ishydrogen, dtype=np.int32)
    n = len(r)
    cap = 64 * 64 * 16
    y, y2 = np.zeros(cap), np.zeros(cap)
    ti, tj = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    ni, nj = C.c_int(0), C.c_int(0)
    rc = lib.agbnp_hip_host_tables(n, _dp(r), _ip(h), C.byref(ni), C.byref(nj), _dp(y), _dp(y2), cap, _ip(ti), _ip(tj))
    if rc != _lib.OK:
        raise OpenMMException(_lib.last_error(None))
    k = ni.value * nj.value * 16
    return dict(y=y[:k].reshape(ni.value, nj.value, 16), y2=y2[:k].reshape(ni.value, nj.value, 16), type_screened=ti,
                type_screener=tj)


class AGBNPContext:
    """Minimal Context + ForceImpl stand-in: positions in, (energy, forces) out, through the kernel."""

    def __init__(self, force, device=0):
        self._force = force
        self._kernel = HipCalcAGBNPForceKernel(device)
        self._kernel.initialize(force)  # AGBNPForceImpl::initialize
        self._positions = None

    @property
    def kernel(self):
        return self._kernel

    def setPositions(self, positions):
        self._positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)

    def getState(self, groups=-1):
        """Returns (potential energy, forces[N,3]).  Honours the force-group mask like
        AGBNPForceImpl::calcForcesAndEnergy (openmmapi/src/AGBNPForceImpl.cpp:32-36)."""
        if self._positions is None:
            raise OpenMMException("Particle positions have not been set")
        forces = np.zeros((self._kernel.numParticles, 3))
        if (groups & (1 << self._force.getForceGroup())) == 0:
            return 0.0, forces
        e = self._kernel.execute(self._positions, forces, True, True)
        return e, forces

    def _update_parameters(self, force):
        self._kernel.copyParametersToContext(force)
