"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle, the committed golden
vectors and size-independent properties.

Tolerance: the north star asks for energies and per-atom forces within 1e-4 kJ/mol (kJ/mol/nm) of the
Reference platform.  The engine computes in FP64, so the tests hold it to TIGHT = 1e-7 (three orders
below the bar; observed differences are ~1e-11) and state the bar next to it."""
import ctypes as C
import os

import numpy as np
import pytest

import openmm_agbnp_plugin_amd as P
from openmm_agbnp_plugin_amd import _lib
from oracle import Oracle

pytestmark = pytest.mark.gpu

BAR = 1e-4    # north-star tolerance, kJ/mol and kJ/mol/nm
TIGHT = 1e-7  # what FP64 on both sides actually delivers, with margin

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_vectors.npz"))


@pytest.fixture()
def six_launches(monkeypatch):
    """The six-launch path (k_prep in front of every evaluation).  Tests of the overflow log that make the geometry JUMP (a
    squeezed protein in the middle of a queue) count on it: in the default five-launch mode a jump beyond the neighbour
    masks' skin voids an evaluation of its own (tests/test_gpu_five_launches.py holds that mode's version of these tests)."""
    monkeypatch.setenv("AGBNP_HIP_FIVE_LAUNCHES", "0")


def gpu_eval(system, version, pos=None, method=P.AGBNPForce.NoCutoff, cutoff=1.0):
    force = P.AGBNPForce.from_arrays(*system.params(), version=version)
    force.setNonbondedMethod(method)
    force.setCutoffDistance(cutoff)
    ctx = P.AGBNPContext(force)
    ctx.setPositions(system.pos if pos is None else pos)
    e, f = ctx.getState()
    return e, f, ctx


def assert_close(e, f, eo, fo, tol=TIGHT):
    assert tol <= BAR
    assert abs(e - eo) < tol * max(1.0, abs(eo) * 1e-3), f"energy differs by {abs(e - eo):.3e}"
    assert np.abs(f - fo).max() < tol, f"forces differ by {np.abs(f - fo).max():.3e}"


# ---- BASELINE.json configs 1-3 -----------------------------------------------------------------------
def test_config1_trpcage_gaussvol_nocutoff(gpu_required, systems):
    s = systems("trpcage")
    e, f, ctx = gpu_eval(s, 0)
    eo, fo = Oracle(*s.params(), version=0).execute(s.pos)
    assert_close(e, f, eo, fo)
    assert abs(e - float(GOLDEN["trpcage_v0_energy"])) < TIGHT
    assert np.abs(f - GOLDEN["trpcage_v0_forces"]).max() < TIGHT


def test_config2_trpcage_agbnp1_cutoff_nonperiodic(gpu_required, systems):
    """CutoffNonPeriodic 1.2 nm is accepted; like the Reference platform it does not change the numbers."""
    s = systems("trpcage")
    e, f, ctx = gpu_eval(s, 1, method=P.AGBNPForce.CutoffNonPeriodic, cutoff=1.2)
    eo, fo = Oracle(*s.params(), version=1).execute(s.pos)
    assert_close(e, f, eo, fo)
    assert np.abs(f - GOLDEN["trpcage_v1_forces"]).max() < TIGHT
    np.testing.assert_allclose(ctx.kernel.vector("born"), GOLDEN["trpcage_v1_born"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("version", [0, 1])
def test_config3_thrombin(gpu_required, systems, version):
    s = systems("1dwc")
    e, f, ctx = gpu_eval(s, version)
    assert abs(e - float(GOLDEN[f"1dwc_v{version}_energy"])) < TIGHT
    assert np.abs(f - GOLDEN[f"1dwc_v{version}_forces"]).max() < TIGHT
    np.testing.assert_allclose(ctx.kernel.vector("selfvol_vdw"), GOLDEN[f"1dwc_v{version}_selfvol_vdw"], rtol=0, atol=1e-14)
    # tree size of the survey: 216146 slots = root + 4152 atoms + overlaps; the engine keeps
    # heavy atoms only: 216146 - 1 - 2068 hydrogens
    assert int(ctx.kernel.scalar("total_nodes")) == 216146 - 1 - (s.n - s.nheavy)
    assert int(ctx.kernel.scalar("max_subtree_nodes")) == 376 + 1


# ---- fast mode: the OpenCL platform's semantics (cutoff on every pair stage), reported separately --------------------
FAST_TOL = 1e-7  # FP64 on both sides, same truncation rule: the same tolerance as the reference mode


@pytest.mark.parametrize("name,cutoff", [("trpcage", 1.2), ("1dwc", 1.0), ("fixture264", 0.8), ("1dwc", 2.5)])
def test_fast_mode_matches_the_cutoff_oracle(gpu_required, systems, name, cutoff):
    """AGBNP_HIP_MODE_FAST against the oracle's cutoff switch (oracle/agbnp_oracle.cpp: restatement of
    AGBNPBornRadii.cl:268,430 / AGBNPGBEnergy.cl:145,186), on the file geometry and a jittered one.  A cutoff beyond the
    tables' 2 nm reach (2.5 nm) only truncates GB."""
    s = systems(name)
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    force.setCutoffDistance(cutoff)
    k = P.HipCalcAGBNPForceKernel(mode="fast")
    k.initialize(force)
    oracle = Oracle(*s.params(), version=1, cutoff=cutoff)
    reference = Oracle(*s.params(), version=1)
    for pos in (s.pos, s.jittered(1, sigma=0.004)):
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo, tol=FAST_TOL)
        er, fr = reference.execute(pos)
        assert abs(e - er) > 1e-3  # and it IS a different model: the truncated pairs are missing
    # back to the reference semantics on the same context
    k.set_mode("reference")
    f = np.zeros((s.n, 3))
    e = k.execute(s.pos, f)
    er, fr = reference.execute(s.pos)
    assert_close(e, f, er, fr)


def test_fast_mode_with_a_huge_cutoff_is_the_reference_mode(gpu_required, systems):
    s = systems("trpcage")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    force.setCutoffDistance(100.0)
    k = P.HipCalcAGBNPForceKernel(mode="fast")
    k.initialize(force)
    f = np.zeros((s.n, 3))
    e = k.execute(s.pos, f)
    eo, fo = Oracle(*s.params(), version=1).execute(s.pos)
    assert_close(e, f, eo, fo)


def test_fast_mode_single_precision_gb(gpu_required, systems):
    """AGBNP_HIP_MODE_FAST | AGBNP_HIP_MODE_SINGLE: GB pair terms in packed FP32 (the precision of the reference's OpenCL
    platform).  Stays within single-precision distance of the FP64 fast mode -- 1e-6 relative in the energy, 2e-2
    kJ/mol/nm in the forces -- and is NOT bit-equal to it; the bit is rejected without the fast mode."""
    s = systems("1dwc")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    force.setCutoffDistance(1.0)
    k64 = P.HipCalcAGBNPForceKernel(mode="fast")
    k64.initialize(force)
    k32 = P.HipCalcAGBNPForceKernel(mode="fast+single")
    k32.initialize(force)
    for pos in (s.pos, s.jittered(3, sigma=0.004)):
        f64, f32 = np.zeros((s.n, 3)), np.zeros((s.n, 3))
        e64, e32 = k64.execute(pos, f64), k32.execute(pos, f32)
        assert e32 != e64
        assert abs(e32 - e64) < 1e-6 * abs(e64) + 1e-2
        assert np.abs(f32 - f64).max() < 2e-2
        assert np.abs(f32 - f64).max() < 1e-4 * np.abs(f64).max()
    lib = _lib.load()
    assert lib.agbnp_hip_set_mode(k32._h, 4) != _lib.OK  # single precision without the fast mode
    assert lib.agbnp_hip_get_mode(k32._h) == 5


@pytest.mark.parametrize("name,evaluations,checked,sigma", [("trpcage", 200, 30, 0.01), ("1dwc", 80, 10, 0.006)])
@pytest.mark.parametrize("version", [0, 1])
def test_soak_with_kicks_follows_the_oracle(gpu_required, systems, name, evaluations, checked, sigma, version):
    """A long sequence of evaluations on ONE context (the forest packing, the work-slot rows and the capacity variant are
    carried from evaluation to evaluation), geometries jittered with a larger kick every 37 steps so that the packing has
    to follow trees that change shape; a random subset is compared with the oracle."""
    s = systems(name)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=version))
    oracle = Oracle(*s.params(), version=version)
    rng = np.random.default_rng(7 + version)
    check = set(rng.choice(evaluations, checked, replace=False).tolist())
    for i in range(evaluations):
        pos = s.jittered(i, sigma=sigma * (3.0 if i % 37 == 0 else 1.0))
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        if i in check:
            eo, fo = oracle.execute(pos)
            assert_close(e, f, eo, fo)


@pytest.mark.parametrize("name", ["trpcage", "1dwc", "2clr"])
def test_work_slot_rows_hold_every_subtree_exactly_once(gpu_required, systems, name):
    """What the bookkeeping (packing in the GB launch, dealing in the chain-rule launch) hands to the next evaluation's
    tree kernel, read back through the diagnostic accessor: every work slot holds 1..8 work items, every subtree appears
    with all of its parts exactly once (a shared subtree as `parts` items, part numbers 0..parts-1), and the number of
    rows is the forest count the engine reports."""
    s = systems(name)
    ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
    lib = _lib.load()
    lib.agbnp_debug_get_packing.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for step in range(4):
        ctx.setPositions(s.jittered(step, sigma=0.004))
        ctx.getState()
        cap = 8 * s.nheavy + 64
        order, start, nf = (C.c_int * cap)(), (C.c_int * cap)(), C.c_int(0)
        assert lib.agbnp_debug_get_packing(ctx.kernel._h, order, cap, start, cap, C.byref(nf), None) == _lib.OK
        nf = nf.value
        start = np.array(start[:nf + 1])
        items = np.array(order[:start[nf]])
        counts = np.diff(start)
        assert nf >= 1 and counts.min() >= 1 and counts.max() <= 8
        subtree, part, parts = items & 0xFFFFFF, (items >> 24) & 3, ((items >> 26) & 3) + 1
        assert part.max() < parts.max() + 1 and (part < parts).all()
        assert sorted(set(subtree.tolist())) == list(range(s.nheavy))   # every subtree is somebody's work
        seen = {}
        for h, p, n in zip(subtree.tolist(), part.tolist(), parts.tolist()):
            seen.setdefault(h, []).append((p, n))
        for h, lst in seen.items():
            n = lst[0][1]
            assert all(x[1] == n for x in lst) and sorted(x[0] for x in lst) == list(range(n)), (h, lst)
        if step > 0:  # (the first evaluation runs one subtree per slot)
            assert nf == int(ctx.kernel.scalar("forests"))


# ---- deterministic mode ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["1dwc", "trpcage"])
def test_deterministic_mode_is_bit_reproducible(gpu_required, systems, name):
    """AGBNP_HIP_MODE_DETERMINISTIC: the same geometries evaluated by two independent contexts, in different sequences
    (so that the forests are packed differently and the atomics land in different orders), give BIT-identical energies
    and forces; the default mode is allowed to differ in the last bits and usually does.  Both stay at the oracle."""
    s = systems(name)
    geoms = [s.pos, s.jittered(1, sigma=0.004), s.jittered(2, sigma=0.004)]

    def run(mode, order):
        k = P.HipCalcAGBNPForceKernel(mode=mode)
        k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
        out = {}
        for g in order:
            f = np.zeros((s.n, 3))
            out[g] = (k.execute(geoms[g], f), f)
        return out

    a = run("deterministic", [0, 1, 2, 0])
    b = run("deterministic", [2, 2, 1, 0])
    for g in range(3):
        assert a[g][0] == b[g][0], f"energy of geometry {g} differs between two runs: {a[g][0]!r} vs {b[g][0]!r}"
        assert np.array_equal(a[g][1], b[g][1]), f"forces of geometry {g} differ by {np.abs(a[g][1] - b[g][1]).max():.3e}"
    oracle = Oracle(*s.params(), version=1)
    for g in range(3):
        eo, fo = oracle.execute(geoms[g])
        assert_close(a[g][0], a[g][1], eo, fo)
    c = run("reference", [0, 1, 2, 0])
    assert abs(c[0][0] - a[0][0]) < 1e-6 and np.abs(c[0][1] - a[0][1]).max() < 1e-7  # the quanta are far below the parity bar


# ---- the reference's own fixture and known answers -------------------------------------------------------
@pytest.mark.parametrize("version", [0, 1])
def test_reference_fixture_known_answers(gpu_required, systems, version):
    from tests.pins import PROBE, REFERENCE_PRINTED
    s = systems("fixture264")
    e, f, ctx = gpu_eval(s, version)
    want = REFERENCE_PRINTED[version]
    sig = lambda x: float(f"{x:.6g}")
    assert sig(e) == want["energy"]
    p = s.pos.copy()
    p[PROBE["atom"], PROBE["direction"]] += PROBE["offset"]
    ctx.setPositions(p)
    e2, _ = ctx.getState()
    assert sig(e2) == want["energy_moved"]
    assert sig(e2 - e) == want["change"]
    assert sig(-f[PROBE["atom"], PROBE["direction"]] * PROBE["offset"]) == want["change_from_gradient"]
    assert np.abs(f - GOLDEN[f"fixture264_v{version}_forces"]).max() < TIGHT


# ---- many geometries -------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,version,steps", [("fixture264", 1, 6), ("trpcage", 0, 4), ("trpcage", 1, 6), ("1dwc", 1, 2)])
def test_jittered_geometries(gpu_required, systems, name, version, steps):
    s = systems(name)
    o = Oracle(*s.params(), version=version)
    force = P.AGBNPForce.from_arrays(*s.params(), version=version)
    ctx = P.AGBNPContext(force)
    for step in range(steps):
        pos = s.jittered(step, sigma=0.004)
        ctx.setPositions(pos)
        e, f = ctx.getState()
        eo, fo = o.execute(pos)
        assert_close(e, f, eo, fo)


def test_second_larger_protein(gpu_required, systems):
    """2clr (5983 atoms): its largest subtree (479 nodes) is beyond the 432-node store that five workgroups per CU
    share.  Round 4: the few subtrees that big are SHARED among several work items (each expands a residue class of the
    level-2 branches) instead of moving every forest of the system to the (512, 64) store: the engine stays on the
    smallest variant -- and stays exact on jittered geometries with the forests on."""
    s = systems("2clr")
    e, f, ctx = gpu_eval(s, 1)
    oracle = Oracle(*s.params(), version=1)
    eo, fo = oracle.execute(s.pos)
    assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("variant")) == 0
    assert int(ctx.kernel.scalar("max_subtree_nodes")) > 432  # (the subtree IS larger than the store)
    for step in range(2):
        pos = s.jittered(step, sigma=0.004)
        ctx.setPositions(pos)
        e, f = ctx.getState()
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
        assert int(ctx.kernel.scalar("forests")) < s.nheavy


def test_mid_size_system_packs_its_forests_into_one_round(gpu_required, systems):
    """2clr's 3358 work items weigh ~1120 stores' worth; the classes' rule dealt them over 2560 forests (two rounds of the
    1280 resident tree workgroups, a third full).  The rounds rule of the bookkeeping (pair_kernels.hip, packing_role)
    packs them into ONE round when the total weight allows it: at most 1280 forests, every subtree built exactly once
    (the tree statistics equal those of the unpacked first evaluation), the oracle's numbers on every step of a short
    queue of jittered geometries, no evaluation withheld."""
    torch = pytest.importorskip("torch")
    s = systems("2clr")
    oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    f0 = np.zeros((s.n, 3))
    k.execute(s.pos, f0)  # unpacked (and the repeat that shares the 479-node subtree)
    total0 = int(k.scalar("total_nodes"))
    f1 = np.zeros((s.n, 3))
    e1 = k.execute(s.pos, f1)  # planned packing
    eo, fo = oracle.execute(s.pos)
    assert_close(e1, f1, eo, fo)
    assert int(k.scalar("total_nodes")) == total0
    assert int(k.scalar("forests")) <= 1280, int(k.scalar("forests"))
    dev = torch.device("cuda:0")
    geoms = [s.jittered(70 + step) for step in range(8)]
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(8):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0, k.withheld()
    assert int(k.scalar("forests")) <= 1280 and int(k.scalar("variant")) == 0
    want = [oracle.execute(g) for g in geoms]
    assert abs(ene.item() - sum(w[0] for w in want)) < 8 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 8 * TIGHT


def test_one_round_packing_in_gaussvol_mode_and_deterministic_mode(gpu_required, systems):
    """The rounds rule where the bookkeeping rides in other launches: version 0 (both halves of the role in k_outputs) and the
    deterministic mode (tile kernels: the role in the GB tile launch, bit-identical results on a second context)."""
    s = systems("2clr")
    for version in (0, 1):
        oracle = Oracle(*s.params(), version=version)
        outs = []
        for ctx_no in range(2 if version == 1 else 1):
            force = P.AGBNPForce.from_arrays(*s.params(), version=version)
            k = P.HipCalcAGBNPForceKernel(device=0, mode="deterministic") if version == 1 else P.HipCalcAGBNPForceKernel(device=0)
            k.initialize(force)
            f = np.zeros((s.n, 3))
            for step in range(3 + ctx_no):  # (the second context plans from another history: other forests, the same bits)
                k.execute(s.jittered(30 + step), f)
            pos = s.jittered(39)
            f = np.zeros((s.n, 3))
            e = k.execute(pos, f)
            assert int(k.scalar("forests")) <= 1280, (version, int(k.scalar("forests")))
            eo, fo = oracle.execute(pos)
            assert_close(e, f, eo, fo)
            outs.append((e, f.copy()))
        if len(outs) == 2:
            assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def test_atom_order_permutation_follows_the_reference_rules(gpu_required, systems):
    """The tree depends on atom order: the engine must follow the oracle for ANY order."""
    s = systems("fixture264")
    rng = np.random.default_rng(11)
    for perm in (np.arange(s.n)[::-1], rng.permutation(s.n)):
        sp = s.permuted(perm)
        e, f, _ = gpu_eval(sp, 1)
        eo, fo = Oracle(*sp.params(), version=1).execute(sp.pos)
        assert_close(e, f, eo, fo)


# ---- data conventions of the boundary ------------------------------------------------------------------
def test_forces_accumulate_and_energy_is_returned(gpu_required, systems):
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    f0 = np.zeros((s.n, 3))
    e0 = k.execute(s.pos, f0)
    f1 = np.full((s.n, 3), -2.5)
    e1 = k.execute(s.pos, f1, includeForces=False, includeEnergy=False)  # flags are ignored, as in the reference
    assert abs(e1 - e0) < 1e-9  # FP64 atomics: summation order may differ in the last bits between runs
    np.testing.assert_allclose(f1 + 2.5, f0, rtol=0, atol=1e-9)


def test_device_resident_entry_point(gpu_required, systems):
    """agbnp_hip_execute_device: FP64 device buffers, forces and energy ADDED in place, asynchronous."""
    torch = pytest.importorskip("torch")
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(s.pos, dtype=torch.float64, device=dev).contiguous()
    frc = torch.full((s.n, 3), 1.5, dtype=torch.float64, device=dev)
    ene = torch.full((1,), 10.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0
    eo, fo = Oracle(*s.params(), version=1).execute(s.pos)
    assert abs((ene.item() - 10.0) / 3 - eo) < TIGHT
    assert np.abs((frc.cpu().numpy() - 1.5) / 3 - fo).max() < TIGHT


@pytest.mark.parametrize("precision", ["double", "mixed", "single"])
def test_openmm_context_data_conventions(gpu_required, systems, precision):
    """agbnp_hip_execute_openmm: posq real4 in a REORDERED atom order with padding (+ the float correction array of
    mixed precision), atomIndex map, 2^32 fixed-point force planes [x | y | z] over the padded count, energy added to
    one slot of the context's accumulator -- the conventions of the reference's OpenCL platform
    (OpenCLAGBNPKernels.cpp:541-556, GVolReduceTree.cl:92-121).  Against the oracle through a random permutation."""
    torch = pytest.importorskip("torch")
    s = systems("trpcage")
    n, padded = s.n, 288  # OpenMM pads the atom count to a multiple of 32
    rng = np.random.default_rng(5)
    atom_index = rng.permutation(n).astype(np.int32)  # context slot -> particle
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    posq64 = np.zeros((padded, 4))
    posq64[:n, :3] = s.pos[atom_index]
    posq64[:n, 3] = s.charge[atom_index]
    if precision == "double":
        posq = torch.tensor(posq64, dtype=torch.float64, device=dev)
        corr = None
        used = posq64[:n, :3]
    else:
        hi = posq64.astype(np.float32)
        posq = torch.tensor(hi, dtype=torch.float32, device=dev)
        lo = (posq64 - hi.astype(np.float64)).astype(np.float32)
        corr = torch.tensor(lo, dtype=torch.float32, device=dev) if precision == "mixed" else None
        used = (hi.astype(np.float64) + (lo.astype(np.float64) if precision == "mixed" else 0.0))[:n, :3]
    geometry = np.zeros((n, 3))
    geometry[atom_index] = used  # the positions the engine actually sees, in particle order
    eo, fo = Oracle(*s.params(), version=1).execute(geometry)
    index = torch.tensor(atom_index, device=dev)
    start = rng.integers(-2 ** 40, 2 ** 40, size=3 * padded)
    fixed = torch.tensor(start, dtype=torch.int64, device=dev)
    energy_is_double = precision != "single"
    ebuf = torch.full((64,), 2.5, dtype=torch.float64 if energy_is_double else torch.float32, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    reps = 2
    for _ in range(reps):
        k.execute_openmm(posq.data_ptr(), precision == "double", corr.data_ptr() if corr is not None else 0, index.data_ptr(), padded,
                         fixed.data_ptr(), ebuf.data_ptr(), energy_is_double, 7, stream)
    assert k.finish(stream) == 0
    got = (fixed.cpu().numpy() - start).reshape(3, padded).astype(np.float64) / 2.0 ** 32 / reps
    assert not got[:, n:].any()  # padding slots untouched
    f_slots = got[:, :n].T       # forces by context slot
    assert np.abs(f_slots - fo[atom_index]).max() < 1e-6  # the fixed point resolves 2^-32 = 2.3e-10 per add
    e = (ebuf.cpu().numpy().astype(np.float64) - 2.5)
    assert not np.delete(e, 7).any()
    assert abs(e[7] / reps - eo) < (TIGHT if energy_is_double else 2e-3)  # a float accumulator holds 7 digits of 2e3 kJ/mol


@pytest.mark.parametrize("adapter_launch", [False, True])
def test_openmm_context_reorders_its_atoms(gpu_required, systems, monkeypatch, adapter_launch):
    """OpenMM reorders its atoms now and then: the atomIndex array keeps its address and changes its contents (and posq
    with it).  agbnp_hip_execute_openmm reads posq through maps of the order it last saw; k_prep checks them against
    atomIndex in every evaluation, so the first evaluation after a reorder is WITHHELD (nothing reaches the context's
    buffers), finish() reports it, the maps are rebuilt and the repeat is right.  With AGBNP_HIP_ADAPTER_LAUNCH=1 (an adapter
    launch per evaluation, rounds 1-2) nothing is withheld."""
    torch = pytest.importorskip("torch")
    if adapter_launch:
        monkeypatch.setenv("AGBNP_HIP_ADAPTER_LAUNCH", "1")
    s = systems("trpcage")
    n, padded = s.n, 288
    rng = np.random.default_rng(9)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    eo, fo = Oracle(*s.params(), version=1).execute(s.pos)
    dev = torch.device("cuda:0")
    index = torch.zeros(padded, dtype=torch.int32, device=dev)
    posq = torch.zeros((padded, 4), dtype=torch.float64, device=dev)
    fixed = torch.zeros(3 * padded, dtype=torch.int64, device=dev)
    ebuf = torch.zeros(8, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def reorder():
        order = rng.permutation(n).astype(np.int32)
        full = np.concatenate([order, np.arange(n, padded, dtype=np.int32)])
        index.copy_(torch.tensor(full))  # same address, new contents
        host = np.zeros((padded, 4))
        host[:n, :3] = s.pos[order]
        posq.copy_(torch.tensor(host))
        return order

    def run():
        k.execute_openmm(posq.data_ptr(), True, 0, index.data_ptr(), padded, fixed.data_ptr(), ebuf.data_ptr(), True, 0, stream)

    def check(order, evaluations):
        torch.cuda.synchronize()
        got = fixed.cpu().numpy().reshape(3, padded).astype(np.float64) / 2.0 ** 32 / evaluations
        assert np.abs(got[:, :n].T - fo[order]).max() < 1e-6
        assert abs(ebuf.cpu().numpy()[0] / evaluations - eo) < TIGHT
        fixed.zero_()
        ebuf.zero_()

    order = reorder()
    torch.cuda.synchronize()
    run()
    run()
    assert k.finish(stream) == 0
    check(order, 2)
    order = reorder()  # OpenMM's reorderAtoms()
    torch.cuda.synchronize()
    run()
    withheld = k.finish(stream)
    if adapter_launch:
        assert withheld == 0
        check(order, 1)
    else:
        assert withheld == 1 and k.withheld() == [0]
        torch.cuda.synchronize()
        assert not fixed.cpu().numpy().any() and not ebuf.cpu().numpy().any()  # nothing of the stale evaluation arrived
    run()
    run()
    assert k.finish(stream) == 0
    check(order, 2)
    order = reorder()
    torch.cuda.synchronize()
    k.atom_order_changed()  # told beforehand (the glue compares the context's host copy of the order): nothing is lost
    run()
    assert k.finish(stream) == 0
    check(order, 1)


def test_evaluation_is_graph_capturable(gpu_required, systems):
    """One evaluation = seven kernel launches on the caller's stream, no host synchronisation, no allocation after
    the first call: it can be captured into a HIP graph and replayed on new positions (MD inner loops)."""
    torch = pytest.importorskip("torch")
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(s.pos, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # warm-up outside the capture: scratch allocation, capacity negotiation
        k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), side.cuda_stream)
        assert k.finish(side.cuda_stream) == 0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        frc.zero_()
        ene.zero_()
        k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), torch.cuda.current_stream().cuda_stream)
    oracle = Oracle(*s.params(), version=1)
    for step in range(3):
        geometry = s.jittered(step)
        pos.copy_(torch.tensor(geometry, dtype=torch.float64))
        graph.replay()
        torch.cuda.synchronize()
        eo, fo = oracle.execute(geometry)
        assert abs(ene.item() - eo) < TIGHT
        assert np.abs(frc.cpu().numpy() - fo).max() < TIGHT


@pytest.mark.parametrize("base_scale,bad_scale,variant_changes", [(1.3, 1.0, False), (1.0, 0.85, True)])
def test_queued_evaluations_withhold_an_overflowed_one(gpu_required, systems, monkeypatch, base_scale, bad_scale, variant_changes, six_launches):
    """Several evaluations are queued on a stream before agbnp_hip_finish; the middle one overflows.
    (1.3, 1.0): the packing is planned on a swollen molecule (largest subtree 58 nodes) and told not to spread the
    forests over the idle workgroups (tuning knob, read when the context is created): eight subtrees per forest; the
    middle geometry is the real one (216 k nodes, largest subtree 377): the packed forests do not fit, every single
    subtree does -- no variant change.
    (1.0, 0.85): a subtree reaches 1977 nodes, two capacity variants up.
    The overflowed evaluation must add NOTHING to the caller's buffers, finish() must name it although later
    evaluations have long reset the per-evaluation status words, and repeating it afterwards must complete the sums."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    scaled = lambda pos, f: centre + f * (pos - centre)
    geoms = [scaled(s.jittered(0), base_scale), scaled(s.jittered(1), base_scale), scaled(s.pos, bad_scale),
             scaled(s.jittered(2), base_scale), scaled(s.jittered(3), base_scale)]
    oracle = Oracle(*s.params(), version=1)
    want = [oracle.execute(g) for g in geoms]
    if not variant_changes:
        monkeypatch.setenv("AGBNP_HIP_ROUND_PERMILLE", "100")
        # (round 6 heals such forests inside the tree launch -- tests/test_gpu_healing.py holds that version of this scenario;
        # AGBNP_HIP_HEAL=0 keeps the withheld-evaluation protocol of rounds 2-5 under test, which capacity overflows still use)
        monkeypatch.setenv("AGBNP_HIP_HEAL", "0")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda i: k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    for i in (0, 1):  # settle: the second evaluation already runs on packed forests
        run(i)
    assert k.finish(stream) == 0
    frc.zero_()
    ene.zero_()
    gen = k.generation()
    for i in range(5):
        run(i)
    assert k.finish(stream) == 1
    assert k.withheld() == [2]
    clean = [0, 1, 3, 4]
    assert abs(ene.item() - sum(want[i][0] for i in clean)) < 4 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(want[i][1] for i in clean)).max() < 4 * TIGHT
    for attempt in range(6):  # one repeat per capacity variant the squeezed trees have to climb (+ one unpacked)
        run(2)
        if k.finish(stream) == 0:
            break
        assert k.withheld() == [0]
    else:
        raise AssertionError("the repeat did not converge")
    assert (k.generation() != gen) == variant_changes  # a captured graph is stale only after a variant change
    assert abs(ene.item() - sum(w[0] for w in want)) < 5 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 5 * TIGHT
    assert k.finish(stream) == 0  # an empty log stays empty


def test_graph_replay_survives_parameter_updates_and_reports_staleness(gpu_required, systems):
    """A captured evaluation keeps working after updateParametersInContext (device copies are rewritten in place) and
    the generation counter tells the caller when a re-capture is due (capacity variant raised by an overflow that a
    replayed step reported through the sticky log)."""
    torch = pytest.importorskip("torch")
    s = systems("fixture264")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(force)
    dev = torch.device("cuda:0")
    pos = torch.tensor(s.pos, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    side = torch.cuda.Stream()

    def capture():
        with torch.cuda.stream(side):  # outside the capture: settle the capacity variant (this fixture's largest
            for _ in range(6):         # subtree, 441 nodes, is beyond the smallest store), allocate the scratch
                k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), side.cuda_stream)
                if k.finish(side.cuda_stream) == 0:
                    break
            else:
                raise AssertionError("capacity negotiation did not converge")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            frc.zero_()
            ene.zero_()
            k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return g, k.generation()

    def replay_and_compare(graph, oracle, geometry):
        pos.copy_(torch.tensor(geometry, dtype=torch.float64))
        graph.replay()
        torch.cuda.synchronize()
        eo, fo = oracle.execute(geometry)
        assert abs(ene.item() - eo) < TIGHT and np.abs(frc.cpu().numpy() - fo).max() < TIGHT

    graph, gen = capture()
    replay_and_compare(graph, Oracle(*s.params(), version=1), s.jittered(0))
    # new charges and alphas: same graph, new numbers
    for i in range(s.n):
        r, g, a, q, h = force.getParticleParameters(i)
        force.setParticleParameters(i, r, g, a * 0.7, q * 1.1, h)
    k.copyParametersToContext(force)
    assert k.generation() == gen
    oracle2 = Oracle(s.radius, s.gamma, s.alpha * 0.7, s.charge * 1.1, s.ishydrogen, version=1)
    replay_and_compare(graph, oracle2, s.jittered(1))
    # a replayed step on a geometry that outgrows the store (5527-node subtree): withheld, logged, variant raised
    centre = s.pos.mean(axis=0)
    squeezed = centre + 0.8 * (s.pos - centre)
    pos.copy_(torch.tensor(squeezed, dtype=torch.float64))
    graph.replay()
    torch.cuda.synchronize()
    assert ene.item() == 0.0 and not frc.cpu().numpy().any()  # zeroed inside the graph, nothing added
    pos.copy_(torch.tensor(s.pos, dtype=torch.float64))
    graph.replay()
    # (replays since the finish inside capture(): two good ones, the squeezed one, one good one)
    assert k.finish(torch.cuda.current_stream().cuda_stream) == 1 and k.withheld() == [2]
    assert k.generation() != gen
    pos.copy_(torch.tensor(squeezed, dtype=torch.float64))
    graph, gen = capture()  # climbs to the variant that holds the squeezed trees, then captures its kernels
    replay_and_compare(graph, oracle2, squeezed)
    replay_and_compare(graph, oracle2, s.pos)


def test_update_parameters_in_context(gpu_required, systems):
    s = systems("trpcage")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    ctx = P.AGBNPContext(force)
    ctx.setPositions(s.pos)
    e0, _ = ctx.getState()
    for i in range(s.n):
        r, g, a, q, h = force.getParticleParameters(i)
        force.setParticleParameters(i, r, g * 1.0, a * 0.5, q * 0.9, h)
    force.updateParametersInContext(ctx)
    e1, f1 = ctx.getState()
    eo, fo = Oracle(s.radius, s.gamma, s.alpha * 0.5, s.charge * 0.9, s.ishydrogen, version=1).execute(s.pos)
    assert e1 != e0
    assert_close(e1, f1, eo, fo)
    # radius changes and heavy->hydrogen flips are refused with the reference's messages
    r, g, a, q, h = force.getParticleParameters(0)
    force.setParticleParameters(0, r + 0.01, g, a, q, h)
    with pytest.raises(P.OpenMMException, match="changing atomic radii"):
        force.updateParametersInContext(ctx)
    force.setParticleParameters(0, r, g, a, q, True)
    with pytest.raises(P.OpenMMException, match="heavy/hydrogen"):
        force.updateParametersInContext(ctx)
    small = P.AGBNPForce.from_arrays(s.radius[:10], s.gamma[:10], s.alpha[:10], s.charge[:10], s.ishydrogen[:10])
    with pytest.raises(P.OpenMMException, match="number of AGBNP particles has changed"):
        small.updateParametersInContext(ctx)


def test_diagnostic_vectors(gpu_required, systems):
    s = systems("fixture264")
    o = Oracle(*s.params(), version=1)
    o.execute(s.pos)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    with pytest.raises(P.OpenMMException, match="no completed evaluation"):
        k.vector("born")
    k.set_diagnostics(True)
    k.execute(s.pos, np.zeros((s.n, 3)))
    for name in ("selfvol_vdw", "selfvol_large", "born", "scale"):
        np.testing.assert_allclose(k.vector(name), o.vector(name), rtol=0, atol=1e-12)
    assert abs(k.scalar("e_vol1") - o.scalar("e_vol1")) < TIGHT and abs(k.scalar("e_vol2") - o.scalar("e_vol2")) < TIGHT
    assert abs(k.scalar("e_atom") + k.scalar("e_gb_pair") - o.scalar("e_gb") - o.scalar("e_vdw")) < TIGHT
    np.testing.assert_array_equal(k.tables()["type_screener"], o.tables()["type_screener"])


def test_force_group_mask(gpu_required, systems):
    s = systems("fixture264")
    force = P.AGBNPForce.from_arrays(*s.params(), version=0)
    force.setForceGroup(3)
    ctx = P.AGBNPContext(force)
    ctx.setPositions(s.pos)
    e, f = ctx.getState(groups=1 << 1)
    assert e == 0.0 and not f.any()
    e, _ = ctx.getState(groups=1 << 3)
    assert e != 0.0


# ---- edge cases ------------------------------------------------------------------------------------------
def test_tiny_and_ragged_systems(gpu_required, systems):
    s = systems("trpcage")
    heavy = np.flatnonzero(s.ishydrogen == 0)
    cases = {
        "one heavy atom": [heavy[0]],
        "one hydrogen": [np.flatnonzero(s.ishydrogen == 1)[0]],
        "two heavy atoms": list(heavy[:2]),
        "63 atoms": list(range(63)),
        "64 atoms": list(range(64)),
        "65 atoms": list(range(65)),
        "hydrogens only": list(np.flatnonzero(s.ishydrogen == 1)[:20]),
        "heavy only": list(heavy[:70]),
    }
    for label, idx in cases.items():
        sub = s.subset(idx)
        for version in (0, 1):
            e, f, _ = gpu_eval(sub, version)
            eo, fo = Oracle(*sub.params(), version=version).execute(sub.pos)
            assert abs(e - eo) < TIGHT and np.abs(f - fo).max() < TIGHT, (label, version)


def test_far_apart_atoms_have_no_overlaps(gpu_required):
    n = 5
    pos = np.arange(n)[:, None] * np.array([[3.0, 0.0, 0.0]])
    sysm = P.AGBNPSystem("line", pos, np.full(n, 0.17), np.full(n, 48.9), np.full(n, -1.0), np.linspace(-0.5, 0.5, n), np.zeros(n, dtype=np.int32))
    e, f, ctx = gpu_eval(sysm, 1)
    eo, fo = Oracle(*sysm.params(), version=1).execute(sysm.pos)
    assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("total_nodes")) == n


def test_capacity_escalation_on_dense_fixture(gpu_required, systems, monkeypatch):
    """platforms/opencl/tests/gaussvol.dat carries radii already enlarged by 0.5 A: subtrees reach 6577 nodes
    and level 8 is populated, far beyond the LDS variants -> the engine must climb to the global-scratch
    variant on its own (the analogue of the reference OpenCL platform's PanicButton/re-init protocol).
    (AGBNP_HIP_SPLIT_FIT=0: without sharing big subtrees, which would stop the climb one variant earlier.)"""
    monkeypatch.setenv("AGBNP_HIP_SPLIT_FIT", "0")
    s = systems("fixture264_ocl")
    for version in (0, 1):
        e, f, ctx = gpu_eval(s, version)
        eo, fo = Oracle(*s.params(), version=version).execute(s.pos)
        assert_close(e, f, eo, fo)
        assert int(ctx.kernel.scalar("variant")) == 4
        assert int(ctx.kernel.scalar("max_subtree_nodes")) == 6576 + 1
        # a second evaluation on the settled variant reproduces the first
        e2, f2 = ctx.getState()
        assert abs(e2 - e) < 1e-9 and np.abs(f2 - f).max() < 1e-9


# ---- full-size properties (HIV-RT stand-in: 2x2x1 lattice of thrombin, 16608 atoms; config 4) -------------------
def test_config4_lattice_properties(gpu_required, systems):
    s = P.lattice(systems("1dwc"), 2, 2, 1, 7.0)
    assert s.n == 16608
    e, f, ctx = gpu_eval(s, 1)
    # no net force or torque on an isolated system
    assert np.abs(f.sum(axis=0)).max() < 1e-7
    torque = np.cross(s.pos - s.pos.mean(axis=0), f).sum(axis=0)
    assert np.abs(torque).max() < 1e-6
    # rigid motion leaves the energy unchanged and rotates the forces
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    ctx.setPositions(s.pos @ R.T + np.array([1.0, -2.0, 0.5]))
    e2, f2 = ctx.getState()
    assert abs(e2 - e) < 1e-6
    assert np.abs(f2 - f @ R.T).max() < 1e-6
    # the cavity term is additive over non-overlapping copies (7 nm apart), the GB term is not
    e1, _, c1 = gpu_eval(systems("1dwc"), 1)
    for name in ("e_vol1", "e_vol2"):
        assert abs(ctx.kernel.scalar(name) - 4 * c1.kernel.scalar(name)) < 1e-6
    assert int(ctx.kernel.scalar("total_nodes")) == 4 * int(c1.kernel.scalar("total_nodes"))


def test_config4_lattice_against_the_oracle_at_full_size(gpu_required, systems):
    """BASELINE.json config 4 at size (16 608 atoms, 8336 heavy: 131 blocks of neighbour masks, more forests than
    resident workgroups, so the work queue of the tree kernels runs, culled Born / chain-rule tiles): energy and every
    force component against the CPU oracle on the file geometry and on a jittered one (the oracle takes ~8 s each)."""
    s = P.lattice(systems("1dwc"), 2, 2, 1, 7.0)
    oracle = Oracle(*s.params(), version=1)
    ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
    for pos in (s.pos, s.jittered(3, sigma=0.004)):
        ctx.setPositions(pos)
        e, f = ctx.getState()  # first evaluation: one subtree per workgroup; second: packed forests
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("forests")) < s.nheavy
    assert int(ctx.kernel.scalar("forests")) > 5 * 256  # more than fit the device at once


@pytest.mark.parametrize("name,shape,pitch", [("1dwc", (4, 1, 1), 5.5), ("2clr", (1, 1, 1), 0.0)])
def test_far_strips_of_the_gb_stage_are_coulomb_to_rounding(gpu_required, systems, monkeypatch, name, shape, pitch):
    """Reference mode on large systems: a GB strip whose blocks are further apart than sqrt(4 * 60 ln2 * Bmax_I * Bmax_J)
    walks a Coulomb-only loop (exp(-d^2 / 4 B_i B_j) < 2^-60 there: the reference's pair term,
    ReferenceAGBNPKernels.cpp:477-499, IS the Coulomb term to FP64 rounding).  An elongated 16 608-atom stand-in (four
    copies of 1dwc in a row, 5.5 nm pitch: 22 nm long, most strips far) and 2clr (a few per cent of its strips) with the
    test forced on: the oracle's numbers at the usual bar, and the same numbers as with the test switched off to 1e-9."""
    base = systems(name)
    s = P.lattice(base, *shape, pitch) if shape != (1, 1, 1) else base
    pos = s.jittered(5, sigma=0.003)
    out = {}
    for far in ("1", "0"):
        monkeypatch.setenv("AGBNP_HIP_GB_FAR", far)
        ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
        ctx.setPositions(s.pos)
        ctx.getState()
        ctx.setPositions(pos)
        out[far] = ctx.getState()
    eo, fo = Oracle(*s.params(), version=1).execute(pos)
    assert_close(out["1"][0], out["1"][1], eo, fo)
    assert abs(out["1"][0] - out["0"][0]) < 1e-9 * max(1.0, abs(eo) * 1e-3) and np.abs(out["1"][1] - out["0"][1]).max() < 1e-9


def test_finite_difference_gradient_on_gpu(gpu_required, systems):
    s = systems("fixture264")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    ctx = P.AGBNPContext(force)
    ctx.setPositions(s.pos)
    _, f = ctx.getState()
    h = 1e-5
    for atom, d in ((121, 1), (1, 0), (200, 2)):
        pp, pm = s.pos.copy(), s.pos.copy()
        pp[atom, d] += h
        pm[atom, d] -= h
        ctx.setPositions(pp)
        ep, _ = ctx.getState()
        ctx.setPositions(pm)
        em, _ = ctx.getState()
        assert abs(-(ep - em) / (2 * h) - f[atom, d]) < 2e-4 * max(1.0, abs(f[atom, d]))


def test_cxx_mirror_reads_like_the_reference_test(gpu_required, tmp_path):
    """The C++ host mirror (cpp/AGBNPForce.h) + C ABI from a C++ program shaped like the reference's own
    TestReferenceAGBNPForce.cpp, on the reference's own fixture, against v0.reference / v1.reference."""
    import subprocess
    from tests.pins import REFERENCE_PRINTED
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "TestHipAGBNPForce")
    libdir = os.path.join(root, "openmm_agbnp_plugin_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "cxx", "TestHipAGBNPForce.cpp"), "-o", exe,
                    os.path.join(libdir, "libagbnp_hip.so"), f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    data = open(os.path.join(root, "openmm_agbnp_plugin_amd", "data", "fixture264.dat")).read()
    for version in (0, 1):
        want = REFERENCE_PRINTED[version]
        out = subprocess.run([exe, str(version), repr(want["energy"]), repr(want["energy_moved"])], input=data, text=True,
                             capture_output=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.split("\n")
        assert lines[0] == f"Energy: {want['energy']:g}"
        assert f"Energy Change: {want['change']:g}" in out.stdout
        assert f"Energy Change from Gradient: {want['change_from_gradient']:g}" in out.stdout
        assert "PASS" in out.stdout


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_clusters(gpu_required, seed):
    """Synthetic compact clusters with many radius types, mixed hydrogens and random charges: exercises type
    tables, dense overlaps (deep trees) and ragged sizes that the protein fixtures do not."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(40, 330))
    # dense packing: points on a jittered cubic lattice with 0.21 nm spacing (denser than a protein interior)
    side = int(np.ceil(n ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = grid * 0.21 + rng.normal(0, 0.02, (n, 3))
    ish = (rng.random(n) < 0.45).astype(np.int32)
    radii_pool = np.array([0.12, 0.145, 0.155, 0.17, 0.175, 0.18, 0.19, 0.2])
    radius = np.where(ish == 1, 0.121, rng.choice(radii_pool, n))
    gamma = np.where(ish == 1, 0.0, 0.117 * 418.4)
    charge = rng.normal(0, 0.4, n)
    from openmm_agbnp_plugin_amd.systems import vdw_alpha_from_radius
    sysm = P.AGBNPSystem(f"cluster{seed}", pos, radius, gamma, vdw_alpha_from_radius(radius), charge, ish)
    for version in (0, 1):
        e, f, ctx = gpu_eval(sysm, version)
        eo, fo = Oracle(*sysm.params(), version=version).execute(sysm.pos)
        assert_close(e, f, eo, fo, tol=1e-6)


def _lattice_cluster(n, spacing, seed, radii, hydrogen_fraction=0.3):
    rng = np.random.default_rng(seed)
    grid = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = grid * spacing + rng.normal(0, 0.02, (n, 3))
    ish = (rng.random(n) < hydrogen_fraction).astype(np.int32)
    radius = np.where(ish == 1, 0.121, rng.choice(radii, n))
    gamma = np.where(ish == 1, 0.0, 0.117 * 418.4)
    from openmm_agbnp_plugin_amd.systems import vdw_alpha_from_radius
    return P.AGBNPSystem(f"cluster{spacing}_{seed}", pos, radius, gamma, vdw_alpha_from_radius(radius), rng.normal(0, 0.4, n), ish)


@pytest.mark.parametrize("n,spacing,seed,variant,min_local_atoms", [
    (150, 0.24, 1, 2, 0),    # largest subtree 751 nodes  -> (1024, 128) LDS variant
    (150, 0.22, 1, 3, 0),    # 1878 nodes                 -> (2048, 256) LDS variant
    (180, 0.18, 2, 4, 65),   # 19425 nodes and a subtree with 68 local atoms (a node with more than 63 younger
                             # siblings: the survivor masks of the expansion span several words) -> HBM-scratch variant
])
def test_every_capacity_variant_is_exact(gpu_required, monkeypatch, n, spacing, seed, variant, min_local_atoms):
    """Denser-than-protein clusters land on the larger tree variants; each must give the oracle's numbers, not
    just be a stepping stone of the capacity negotiation.  (AGBNP_HIP_SPLIT_FIT=0: every subtree whole, so that each
    variant's kernels are the ones that run; the sharing of big subtrees has the test below.)"""
    monkeypatch.setenv("AGBNP_HIP_SPLIT_FIT", "0")
    sysm = _lattice_cluster(n, spacing, seed, [0.17, 0.18, 0.19, 0.2])
    e, f, ctx = gpu_eval(sysm, 1)
    eo, fo = Oracle(*sysm.params(), version=1).execute(sysm.pos)
    assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("variant")) == variant
    assert int(ctx.kernel.scalar("max_local_atoms")) >= min_local_atoms
    # (round 6: the five-launch mode holds on every LDS store; only the 32 768-node store in HBM goes back to six launches)
    assert int(ctx.kernel.scalar("launches")) == (5 if variant <= 3 else 6)
    # a second, jittered evaluation on the variant the first one settled on (in the mode: its own kernels, its own masks)
    pos2 = sysm.pos + np.random.default_rng(seed + 100).normal(0.0, 0.001, sysm.pos.shape)
    ctx.setPositions(pos2)
    e2, f2 = ctx.getState()
    eo2, fo2 = Oracle(*sysm.params(), version=1).execute(pos2)
    assert_close(e2, f2, eo2, fo2)
    ctx.setPositions(sysm.pos)
    ctx.getState()
    # same tree as the reference: heavy level-1 nodes + everything below (the oracle's level 1 also lists hydrogens)
    o = Oracle(*sysm.params(), version=0)
    o.execute(sysm.pos)
    nheavy = int(np.sum(sysm.ishydrogen == 0))
    assert int(ctx.kernel.scalar("total_nodes")) == nheavy + sum(o.tree_stats()["level_counts"][2:])


def test_big_subtrees_are_shared_in_gaussvol_mode_too(gpu_required):
    """Version 0 (GaussVol only: three launches, the bookkeeping rides in k_outputs) walks the same sharing protocol."""
    sysm = _lattice_cluster(150, 0.24, 1, [0.17, 0.18, 0.19, 0.2])
    oracle = Oracle(*sysm.params(), version=0)
    e, f, ctx = gpu_eval(sysm, 0)
    eo, fo = oracle.execute(sysm.pos)
    assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("variant")) < 2 and int(ctx.kernel.scalar("max_subtree_nodes")) > 512
    for step in range(4):  # planned packings with shared subtrees, evaluation after evaluation
        pos = sysm.pos + np.random.default_rng(step).normal(0.0, 0.001, sysm.pos.shape)
        ctx.setPositions(pos)
        e, f = ctx.getState()
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)


@pytest.mark.parametrize("n,spacing,seed,whole_variant", [(150, 0.24, 1, 2), (150, 0.22, 1, 3)])
def test_big_subtrees_are_shared_before_the_variant_is_raised(gpu_required, n, spacing, seed, whole_variant):
    """The same clusters with the default settings: a lone subtree that outgrows the store is first shared four ways (the
    repeat runs on the four-way fallback packing, later evaluations on parts planned from the measured shapes); only what
    does not fit even then moves the system to a larger store.  Exact either way, on a smaller variant than whole subtrees
    need, and steady: later evaluations are not withheld again."""
    sysm = _lattice_cluster(n, spacing, seed, [0.17, 0.18, 0.19, 0.2])
    oracle = Oracle(*sysm.params(), version=1)
    e, f, ctx = gpu_eval(sysm, 1)
    eo, fo = oracle.execute(sysm.pos)
    assert_close(e, f, eo, fo)
    assert int(ctx.kernel.scalar("variant")) < whole_variant
    k = ctx.kernel
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    geoms = [sysm.pos + np.random.default_rng(step).normal(0.0, 0.001, sysm.pos.shape) for step in range(6)]
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((sysm.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(6):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0, k.withheld()  # planned packings (shared subtrees, forests) hold from here on
    want = [oracle.execute(g) for g in geoms]
    assert abs(ene.item() - sum(w[0] for w in want)) < 6 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 6 * TIGHT


def test_a_lone_subtree_beyond_the_store_heals_on_the_device_while_evaluations_are_queued(gpu_required, systems, monkeypatch):
    """(AGBNP_HIP_HEAL=0: the protocol of rounds 4-5, which remains for what the tree launch cannot heal by itself -- a
    three-way share, a waiting list that is full; with healing on this scenario withholds NOTHING:
    tests/test_gpu_healing.py::test_a_lone_subtree_beyond_the_store_is_built_in_four_parts_at_once.)
    ADVICE r04: a FRESH context whose first evaluations are all queued on the device-resident path before anybody
    reads the log (a captured MD graph, bench.py's drift chunks).  2clr has subtrees of up to 479 nodes; the smallest store
    holds 432, so the first evaluation (one whole subtree per work slot) is withheld.  The device must react by itself: the
    overflowing item records its subtree as too big, that evaluation's bookkeeping hands it to four work items, and
    everything queued behind is COMPLETE -- not withheld until a host finish() arrives -- without the packing's assumed
    capacity being tightened (a lone item is no misprediction of the packing)."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("AGBNP_HIP_HEAL", "0")
    s = systems("2clr")
    oracle = Oracle(*s.params(), version=1)
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(force)
    dev = torch.device("cuda:0")
    geoms = [s.jittered(40 + step) for step in range(6)]
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(6):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 1 and list(k.withheld()) == [0]  # only the evaluation that met the unshared subtree
    assert int(k.scalar("variant")) == 0 and int(k.scalar("pack_level")) == 0
    assert int(k.scalar("max_subtree_nodes")) > 432
    want = [oracle.execute(g) for g in geoms]
    assert abs(ene.item() - sum(w[0] for w in want[1:])) < 5 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want[1:])).max() < 5 * TIGHT
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # the repeat completes the sum
    assert k.finish(stream) == 0
    assert abs(ene.item() - sum(w[0] for w in want)) < 6 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 6 * TIGHT
    # ... and a long queue behind a late finish does not leave the packing switched off: level stays 0, forests are packed
    for rep in range(3):
        for i in range(6):
            k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0
    nheavy = int(np.sum(s.ishydrogen == 0))
    assert int(k.scalar("pack_level")) == 0 and int(k.scalar("forests")) < nheavy


def test_forest_packing_and_its_overflow_protocol(gpu_required, systems):
    """From the second evaluation on, the tree kernels build several subtrees per workgroup, packed from the previous
    evaluation's subtree shapes.  Steady state must stay exact; a geometry that makes the trees grow far beyond the
    prediction (here: the same atoms 20 % closer together) overflows the packed forests and must be repeated
    unpacked -- silently for execute_host -- with the packing tightened afterwards."""
    s = systems("1dwc")  # 2084 subtrees > 1024 resident workgroups: packing pays (a small system keeps one subtree per slot)
    ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    nheavy = int(np.sum(s.ishydrogen == 0))
    for step in range(3):  # steady state on slightly different geometries
        pos = s.jittered(step)
        ctx.setPositions(pos)
        e, f = ctx.getState()
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
        if step > 0:
            assert int(ctx.kernel.scalar("forests")) < nheavy  # packed: fewer work slots than subtrees
    centre = s.pos.mean(axis=0)
    squeezed = centre + 0.85 * (s.pos - centre)
    ctx.setPositions(squeezed)
    e, f = ctx.getState()
    eo, fo = oracle.execute(squeezed)
    assert_close(e, f, eo, fo, tol=1e-6)
    ctx.setPositions(s.pos)  # and back again
    e, f = ctx.getState()
    eo, fo = oracle.execute(s.pos)
    assert_close(e, f, eo, fo)


def test_shared_subtrees_on_a_roomy_device(gpu_required, systems):
    """A small system leaves most of the device idle, so from the second evaluation on its big subtrees are shared by
    several workgroups (each expands a residue class of the level-2 nodes): more work slots than subtrees, same
    numbers, same tree statistics."""
    for name in ("trpcage", "fixture264"):
        s = systems(name)
        nheavy = int(np.sum(s.ishydrogen == 0))
        for version in (0, 1):
            ctx = P.AGBNPContext(P.AGBNPForce.from_arrays(*s.params(), version=version))
            oracle = Oracle(*s.params(), version=version)
            totals = []
            for step in range(4):
                pos = s.jittered(step) if step < 3 else s.pos
                ctx.setPositions(pos)
                e, f = ctx.getState()
                eo, fo = oracle.execute(pos)
                assert_close(e, f, eo, fo)
                totals.append(int(ctx.kernel.scalar("total_nodes")))
                nodes = ctx.kernel.vector("subtree_nodes")
                assert int(nodes.sum()) == totals[-1]
                assert totals[-1] == nheavy + sum(oracle.tree_stats()["level_counts"][2:])
            assert int(ctx.kernel.scalar("forests")) > nheavy


def test_many_radius_types_spill_the_default_lds_allowance(gpu_required):
    """15 distinct heavy radii + the hydrogen radius: 16 x 15 type pairs x 16 knots x 16 B = 61 KB of spline tables,
    beyond the default dynamic-LDS allowance of a workgroup (the launchers must raise it) but inside the budget."""
    radii = list(np.round(np.linspace(0.150, 0.206, 15), 4))
    sysm = _lattice_cluster(200, 0.27, 5, radii, hydrogen_fraction=0.4)
    assert len(set(np.round(sysm.radius, 4))) == 16
    e, f, ctx = gpu_eval(sysm, 1)
    eo, fo = Oracle(*sysm.params(), version=1).execute(sysm.pos)
    assert_close(e, f, eo, fo)


def test_absurd_density_is_a_clean_capacity_error(gpu_required):
    """Atoms packed far beyond any physical density make the overlap tree explode (> 32768 nodes under one
    atom): the engine must climb through its variants and then fail with a message, not crash or hang."""
    rng = np.random.default_rng(3)
    n = 216
    grid = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pos = grid * 0.10 + rng.normal(0, 0.01, (n, 3))
    sysm = P.AGBNPSystem("blob", pos, np.full(n, 0.2), np.full(n, 48.9), np.full(n, -1.0), np.zeros(n), np.zeros(n, dtype=np.int32))
    with pytest.raises(P.OpenMMException, match="exceeds the largest supported capacity"):
        gpu_eval(sysm, 0)


# ---- row form of the range-limited pair stages (neighbour rows with a skin) ------------------------------------------
def test_neighbour_rows_follow_moving_atoms(gpu_required, systems, monkeypatch):
    """Reference mode runs the Born sums and the chain rule over neighbour rows built with a skin (default 0.1 nm) and
    rebuilt on the device once an atom is further than half of it from where it was at the last build.  A walk whose
    steps are small (no rebuild for a while), then larger than the skin (rebuild at once), then small again must match
    the oracle at every step, and the rows must have been rebuilt when -- and only when -- an atom had moved too far."""
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    monkeypatch.setenv("AGBNP_HIP_ROW_SLICE", "256")  # (a fixed slice length: no extra build while it is being tuned)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    rng = np.random.default_rng(11)
    pos = s.pos.copy()
    builds = []
    for step, sigma in enumerate([0.0, 0.003, 0.003, 0.003, 0.08, 0.002, 0.002, 0.03, 0.03, 0.001]):
        pos = pos + rng.normal(0.0, sigma, pos.shape) if sigma > 0 else pos
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
        assert k.scalar("rows_on") == 1
        builds.append(int(k.scalar("row_builds")))
    # sigma 0.003 nm never carries an atom 0.05 nm away in three steps; sigma 0.08 nm does at once; the drift of two
    # steps of 0.03 nm does too
    assert builds[0] == 1 and builds[3] == 1, builds
    assert builds[4] == 2 and builds[6] == 2, builds
    assert builds[-1] >= 3, builds


def test_neighbour_row_overflow_falls_back_to_the_tiles(gpu_required, systems, monkeypatch):
    """A neighbour row that outgrows its stride (forced here: 128 entries) withholds the evaluation like any other
    capacity overflow; the host repeats it on the tile kernels and stays there."""
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    monkeypatch.setenv("AGBNP_HIP_ROW_STRIDE", "128")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    monkeypatch.delenv("AGBNP_HIP_ROW_STRIDE")
    oracle = Oracle(*s.params(), version=1)
    for step in range(3):
        pos = s.jittered(step)
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
        assert k.scalar("rows_on") == 0


@pytest.mark.parametrize("skin", ["0.0", "0.02", "0.3"])
def test_neighbour_rows_are_exact_for_any_skin(gpu_required, systems, monkeypatch, skin):
    """The skin only decides how often the rows are rebuilt (0: at every new geometry), never which pairs are summed."""
    s = systems("fixture264")
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    monkeypatch.setenv("AGBNP_HIP_SKIN", skin)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    for step in range(4):
        pos = s.jittered(step, sigma=0.004)
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
    assert k.scalar("rows_on") == 1
    builds = int(k.scalar("row_builds"))  # (jitter of 0.004 nm: 0.01 nm = half of 0.02 is exceeded by some atom, 0.15 nm never)
    assert builds == 4 if skin == "0.0" else 1 <= builds <= 4 if skin == "0.02" else builds == 1


# ---- round-3 contract fixes ----------------------------------------------------------------------------------------
def test_fast_mode_honours_the_nonbonded_method(gpu_required, systems):
    """The OpenCL platform only has a cutoff for a nonbonded method other than NoCutoff (USE_CUTOFF,
    OpenCLAGBNPKernels.cpp:487,1149-1150): a default force (NoCutoff, cutoff distance 1.0 'without effect') in fast mode
    computes ALL pairs -- the reference mode's numbers -- and CutoffPeriodic is refused (no box crosses the boundary)."""
    s = systems("trpcage")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)  # NoCutoff, cutoff distance 1.0: the reference's defaults
    assert force.getNonbondedMethod() == P.AGBNPForce.NoCutoff and force.getCutoffDistance() == 1.0
    k = P.HipCalcAGBNPForceKernel(mode="fast")
    k.initialize(force)
    f = np.zeros((s.n, 3))
    e = k.execute(s.pos, f)
    eo, fo = Oracle(*s.params(), version=1, cutoff=1.0, method=0).execute(s.pos)  # (the oracle's switch takes the same gate)
    er, fr = Oracle(*s.params(), version=1).execute(s.pos)
    assert eo == er
    assert_close(e, f, er, fr)
    force.setNonbondedMethod(P.AGBNPForce.CutoffPeriodic)
    k2 = P.HipCalcAGBNPForceKernel(mode="fast")
    with pytest.raises(P.OpenMMException, match="CutoffPeriodic"):
        k2.initialize(force)
    k3 = P.HipCalcAGBNPForceKernel()  # reference mode: every method is accepted and inert, as on the Reference platform
    k3.initialize(force)
    f3 = np.zeros((s.n, 3))
    assert_close(k3.execute(s.pos, f3), f3, er, fr)


def test_host_call_does_not_swallow_the_device_log(gpu_required, systems):
    """agbnp_hip_execute_host reads and clears the sticky overflow log for its own repeat protocol.  Evaluations that the
    caller has enqueued with execute_device and not yet finished must still be reported by the caller's next finish()."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    squeezed = centre + 0.85 * (s.pos - centre)  # outgrows the smallest capacity variant
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack([s.pos, squeezed]), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # 0: complete
    k.execute_device(pos[1].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # 1: withheld
    f = np.zeros((s.n, 3))
    e = k.execute(s.jittered(3), f)  # a host call in between: harvests, repeats itself as needed, returns complete numbers
    eo, fo = Oracle(*s.params(), version=1).execute(s.jittered(3))
    assert_close(e, f, eo, fo)
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # 2: complete
    assert k.finish(stream) == 1
    assert k.withheld() == [1]
    assert k.finish(stream) == 0 and k.withheld() == []


def test_packing_back_off_relaxes_again(gpu_required, systems, monkeypatch):
    """Every overflow of a packed forest that is NOT healed inside the tree launch tightens the capacity the packing assumes
    (pack_level + 1); clean evaluations in a row (64, doubled by every tightening) give one step back, so occasional mispredictions do not push a long run to one subtree per slot for good.
    After an overflow the very next clean evaluation plans anew (the unpacked fallback is not kept for a replan period)."""
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_REPLAN_EVERY", "1")
    monkeypatch.setenv("AGBNP_HIP_HEAL", "0")  # (with healing on such forests are built again in smaller sets and the level stays 0)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    centre = s.pos.mean(axis=0)
    f = np.zeros((s.n, 3))
    for step in range(3):
        k.execute(s.jittered(step), f)
    assert k.scalar("pack_level") == 0
    packed = int(k.scalar("forests"))
    assert packed < s.nheavy
    k.execute(centre + 0.9 * (s.pos - centre), f)  # trees grow beyond the plan: the packed forests overflow, the call repeats itself
    level = int(k.scalar("pack_level"))
    assert level >= 1
    k.execute(s.jittered(4), f)
    k.execute(s.jittered(5), f)
    assert int(k.scalar("forests")) < s.nheavy  # packed again at once (tighter than before)
    # (round 6: the level has a memory -- clean EVALUATIONS are counted, 64 of them at least, and every tightening doubles what is
    # asked for before a step is given back: 128 after one tightening, 256 after two)
    for step in range(40):
        k.execute(s.jittered(6 + step), f)
    assert int(k.scalar("pack_level")) == level  # not yet: round 5 would have given a step back after four plans
    for step in range(256 * level + 8):
        k.execute(s.jittered(46 + step % 50), f)
    assert k.scalar("pack_level") == 0
    assert int(k.scalar("forests")) <= packed + 8  # back at the original packing density


def test_poll_reports_the_log_without_synchronising(gpu_required, systems, six_launches):
    """agbnp_hip_poll reads pinned host memory that the device writes at the end of every evaluation: evaluations completed
    since the last finish() and how many of them were withheld -- the same numbers finish() then returns."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack([s.pos, centre + 0.85 * (s.pos - centre)]), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    assert k.poll() == (0, 0)
    for i in (0, 0, 1, 0):  # the third one outgrows the smallest capacity variant
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    torch.cuda.synchronize()  # (the test's own: poll itself never waits)
    assert k.poll() == (4, 1)
    assert k.finish(stream) == 1 and k.withheld() == [2]
    assert k.poll() == (0, 0)


def test_host_entry_short_cut_keeps_the_log_of_the_device_entry_points(gpu_required, systems, six_launches):
    """agbnp_hip_execute_host skips the reads of the device for an evaluation that the pinned status words call complete and
    leaves the overflow log running; none of that may show through the device-resident protocol: poll / wait_verdict do not
    count the host path's evaluations, diagnostics asked for afterwards are those of the last evaluation, the indices that
    finish() reports start at the caller's own first evaluation, and a withheld host evaluation is still repeated."""
    torch = pytest.importorskip("torch")
    from oracle import Oracle
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    squeezed = centre + 0.85 * (s.pos - centre)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    o = Oracle(*s.params(), version=1)
    f = np.zeros((s.n, 3))
    for step in range(3):  # the short cut (after the first, which settles the packing)
        f[:] = 0.0
        e = k.execute(s.jittered(step), f)
    eo, fo = o.execute(s.jittered(2))
    assert_close(e, f, eo, fo)
    assert k.poll() == (0, 0) and k.wait_verdict() == (0, 0)
    assert k.scalar("total_nodes") > 200000  # a diagnostic of the LAST evaluation: caught up with on demand
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack([s.pos, squeezed]), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    f[:] = 0.0
    k.execute(s.jittered(3), f)  # host path again, then straight into the device path
    for i in (0, 1, 0):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.wait_verdict() == (3, 1)
    assert k.finish(stream) == 1 and k.withheld() == [1]
    # a host evaluation that overflows (the squeezed geometry on a fresh context) is repeated inside the call
    k2 = P.HipCalcAGBNPForceKernel()
    k2.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    f[:] = 0.0
    k2.execute(s.pos, f)
    f[:] = 0.0
    e = k2.execute(squeezed, f)
    eo, fo = o.execute(squeezed)
    assert_close(e, f, eo, fo)


def test_entry_points_mixed_at_random_keep_their_books(gpu_required, systems, monkeypatch):
    """A seeded random walk over the entry points of ONE context -- host-buffer evaluations (short cut or repeat inside the
    call), batches of queued device-resident evaluations judged by wait_verdict and read by finish, diagnostics in between
    -- on geometries that alternate between a swollen molecule and the real one, so that the forest packing planned on one
    overflows on the other now and then.  Every host call must return the oracle's numbers; wait_verdict, finish and
    withheld() must agree on how many and which evaluations of a batch were withheld; repeating exactly those must make the
    device buffers the oracle's sum over everything that was asked for."""
    torch = pytest.importorskip("torch")
    from oracle import Oracle
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    pool = [centre + 1.3 * (s.jittered(j) - centre) for j in range(4)] + [s.jittered(10 + j) for j in range(4)]
    o = Oracle(*s.params(), version=1)
    want = [o.execute(g) for g in pool]
    monkeypatch.setenv("AGBNP_HIP_ROUND_PERMILLE", "100")  # (eight subtrees per forest: a packing that mispredicts easily)
    monkeypatch.setenv("AGBNP_HIP_REPLAN_EVERY", "1")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack(pool), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20261004)
    exp_e, exp_f = 0.0, np.zeros((s.n, 3))
    asked = withheld_total = host_calls = 0
    f = np.zeros((s.n, 3))
    for round_ in range(60):
        if round_ >= 6 and rng.random() < 0.4:
            g = int(rng.integers(len(pool)))
            f[:] = 0.0
            e = k.execute(pool[g], f)
            assert_close(e, f, *want[g])
            assert k.poll() == (0, 0)
            host_calls += 1
        else:
            batch = [int(g) for g in rng.integers(len(pool), size=int(rng.integers(1, 6)))]
            if round_ < 6:  # the first batches cross from the swollen molecule to the real one for certain
                batch = [int(rng.integers(4)), int(rng.integers(4)), 4 + int(rng.integers(4)), int(rng.integers(4))]
            for g in batch:
                k.execute_device(pos[g].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
            done, bad = k.wait_verdict()
            assert done == len(batch)
            assert k.finish(stream) == bad
            idx = k.withheld()
            assert len(idx) == bad and all(0 <= i < len(batch) for i in idx) and idx == sorted(set(idx))
            asked += len(batch)
            withheld_total += bad
            for i, g in enumerate(batch):
                if i not in idx:
                    exp_e += want[g][0]
                    exp_f += want[g][1]
            for i in idx:  # exactly those, until they are in
                for attempt in range(8):
                    k.execute_device(pos[batch[i]].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
                    if k.finish(stream) == 0:
                        break
                else:
                    raise AssertionError("a withheld evaluation did not complete in eight repeats")
                exp_e += want[batch[i]][0]
                exp_f += want[batch[i]][1]
        if rng.random() < 0.15:
            assert k.scalar("total_nodes") > 20000  # (a diagnostic: the reads that the host path's short cut put off)
    torch.cuda.synchronize()
    assert host_calls > 10 and asked > 60
    assert withheld_total >= 1, "the walk was meant to cross at least one overflow"
    assert abs(ene.item() - exp_e) < 1e-7 * max(1.0, abs(exp_e) * 1e-3) * 10
    assert np.abs(frc.cpu().numpy() - exp_f).max() < 1e-7 * 10


def test_host_facing_paths_without_pinned_staging(gpu_required, systems, monkeypatch):
    """The host-buffer entry point, finish() and update_parameters() fall back to pageable transfers and blocking reads when
    pinned staging cannot be had (AGBNP_HIP_NO_PINNED_STAGING forces it): same numbers, same overflow protocol."""
    from oracle import Oracle
    monkeypatch.setenv("AGBNP_HIP_NO_PINNED_STAGING", "1")
    s = systems("trpcage")
    centre = s.pos.mean(axis=0)
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(force)
    o = Oracle(*s.params(), version=1)
    f = np.zeros((s.n, 3))
    for geometry in (s.pos, s.jittered(1), centre + 0.7 * (s.pos - centre), s.jittered(2)):  # (the third outgrows the smallest store)
        f[:] = 0.0
        e = k.execute(geometry, f)
        assert_close(e, f, *o.execute(geometry))
    r, g, a, q, h = s.params()
    force2 = P.AGBNPForce.from_arrays(r, g, a, 0.5 * np.asarray(q), h, version=1)
    k.copyParametersToContext(force2)
    f[:] = 0.0
    e = k.execute(s.pos, f)
    assert_close(e, f, *Oracle(r, g, a, 0.5 * np.asarray(q), h, version=1).execute(s.pos))


def test_wait_verdict_judges_every_evaluation_without_draining_the_stream(gpu_required, systems):
    """agbnp_hip_wait_verdict blocks the HOST until the device has written its verdict on every evaluation enqueued since
    the last finish() -- no synchronisation call of ours in between -- and says how many were withheld: the strict
    per-step check of the reference's GPU platform (OpenCLAGBNPKernels.cpp:3599-3634) without its pipeline drain.  What a
    clean verdict promises is checked afterwards: the buffers hold the sum of the complete evaluations."""
    torch = pytest.importorskip("torch")
    from oracle import Oracle
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    dev = torch.device("cuda:0")
    geoms = [s.pos, s.jittered(1), centre + 0.85 * (s.pos - centre)]
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    assert k.wait_verdict() == (0, 0)  # nothing enqueued: nothing to wait for
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.wait_verdict() == (1, 0)
    k.execute_device(pos[1].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.wait_verdict() == (2, 0)
    k.execute_device(pos[2].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # outgrows the smallest capacity variant
    assert k.wait_verdict() == (3, 1)
    with pytest.raises(P.OpenMMException):  # a count the device will not reach: times out, says how far it got
        k.wait_verdict(evaluations=5, timeout=0.05)
    assert k.finish(stream) == 1 and k.withheld() == [2]
    assert k.wait_verdict() == (0, 0)
    o = Oracle(*s.params(), version=1)
    (e0, f0), (e1, f1) = o.execute(geoms[0]), o.execute(geoms[1])
    assert_close(float(ene.cpu()[0]), frc.cpu().numpy(), e0 + e1, f0 + f1)


@pytest.mark.parametrize("name,cutoff", [("2clr", 1.0), ("1dwc_x4", 1.2)])
def test_fast_mode_rows_on_larger_systems(gpu_required, systems, name, cutoff):
    """Fast mode runs all three pair stages in row form (neighbour lists within cutoff + skin, no pair beyond the cutoff is
    met): the second protein and the 16 608-atom lattice (BASELINE config 4) against the oracle's cutoff switch, on the
    file geometry and after the atoms have moved (lists rebuilt)."""
    s = P.lattice(systems("1dwc"), 2, 2, 1, 7.0) if name == "1dwc_x4" else systems(name)
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    force.setCutoffDistance(cutoff)
    k = P.HipCalcAGBNPForceKernel(mode="fast")
    k.initialize(force)
    oracle = Oracle(*s.params(), version=1, cutoff=cutoff)
    rng = np.random.default_rng(5)
    for pos in (s.pos, s.pos + rng.normal(0.0, 0.04, s.pos.shape)):
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo, tol=FAST_TOL)
    assert k.scalar("rows_on") == 1 and int(k.scalar("row_builds")) == 2


def test_neighbour_list_walk_widens_on_demand(gpu_required, systems, monkeypatch):
    """The row launches walk as much of a list as protein density can fill at the current reach; a list that outgrows that
    (forced here with an absurdly low density bound) withholds the evaluation like any capacity overflow, the host doubles
    the walk and repeats: exact numbers, and the row form stays on."""
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    monkeypatch.setenv("AGBNP_HIP_ROW_FILL", "0.05")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    for step in range(2):
        pos = s.jittered(step)
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
    assert k.scalar("rows_on") == 1


def test_slice_length_of_the_neighbour_rows_is_tuned_on_the_device(gpu_required, systems, monkeypatch):
    """A wave of a row launch walks one slice of a list.  When the working workgroups of a launch are a few more than two
    per CU (1dwc with slices of 256 entries: ~600 on 256 CUs) the evaluation that has built the lists asks for one more
    build with longer slices: same numbers whatever the slice length, one extra build per step of 64 entries, and a
    small system (whose launches are far from filling the device) or a fixed AGBNP_HIP_ROW_SLICE is left alone."""
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    s = systems("1dwc")
    oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    for step in range(4):
        pos = s.jittered(step)
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
    tuned = int(k.scalar("row_slice"))
    assert 256 < tuned <= 512 and tuned % 64 == 0
    assert int(k.scalar("row_builds")) == 1 + (tuned - 256) // 64
    for fixed in (256, 448):
        monkeypatch.setenv("AGBNP_HIP_ROW_SLICE", str(fixed))
        k2 = P.HipCalcAGBNPForceKernel()
        k2.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
        for step in range(3):
            pos = s.jittered(step)
            f = np.zeros((s.n, 3))
            e = k2.execute(pos, f)
            eo, fo = oracle.execute(pos)
            assert_close(e, f, eo, fo)
        assert int(k2.scalar("row_slice")) == fixed and int(k2.scalar("row_builds")) == 1
    monkeypatch.delenv("AGBNP_HIP_ROW_SLICE")
    t = systems("trpcage")
    k3 = P.HipCalcAGBNPForceKernel()
    k3.initialize(P.AGBNPForce.from_arrays(*t.params(), version=1))
    to = Oracle(*t.params(), version=1)
    for step in range(3):
        pos = t.jittered(step)
        f = np.zeros((t.n, 3))
        e = k3.execute(pos, f)
        eo, fo = to.execute(pos)
        assert_close(e, f, eo, fo)
    assert int(k3.scalar("row_slice")) == 256 and int(k3.scalar("row_builds")) == 1


def test_two_contexts_on_two_streams_do_not_stall_or_disturb_each_other(gpu_required, systems):
    """Multi-walker use: two contexts enqueue on streams of their own; parameters of one are updated (in place, behind a
    drain of ITS streams only) while evaluations of the other are in flight.  Both end with the oracle's numbers."""
    torch = pytest.importorskip("torch")
    sa, sb = systems("trpcage"), systems("fixture264")
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    forces = [P.AGBNPForce.from_arrays(*sa.params(), version=1), P.AGBNPForce.from_arrays(*sb.params(), version=1)]
    ks = [P.HipCalcAGBNPForceKernel(), P.HipCalcAGBNPForceKernel()]
    for k, f in zip(ks, forces):
        k.initialize(f)
    geoms = [[s.jittered(i) for i in range(6)] for s in (sa, sb)]
    pos = [torch.tensor(np.stack(g), dtype=torch.float64, device=dev).contiguous() for g in geoms]
    frc = [torch.zeros((s.n, 3), dtype=torch.float64, device=dev) for s in (sa, sb)]
    ene = [torch.zeros((1,), dtype=torch.float64, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for w in (0, 1):  # settle capacities (the 264-atom fixture needs the second variant)
        for attempt in range(4):
            ks[w].execute_device(pos[w][0].data_ptr(), frc[w].data_ptr(), ene[w].data_ptr(), streams[w].cuda_stream)
            if ks[w].finish(streams[w].cuda_stream) == 0:
                break
        frc[w].zero_()
        ene[w].zero_()
    torch.cuda.synchronize()
    # walker 1 keeps evaluating; walker 0 evaluates, changes its charges, evaluates again
    for i in range(3):
        ks[1].execute_device(pos[1][i].data_ptr(), frc[1].data_ptr(), ene[1].data_ptr(), streams[1].cuda_stream)
        ks[0].execute_device(pos[0][i].data_ptr(), frc[0].data_ptr(), ene[0].data_ptr(), streams[0].cuda_stream)
    r, g, a, q, h = sa.params()
    q2 = 0.5 * np.asarray(q)
    f2 = P.AGBNPForce.from_arrays(r, g, a, q2, h, version=1)
    assert ks[0].finish(streams[0].cuda_stream) == 0
    ks[0].copyParametersToContext(f2)  # (walker 1's evaluations may still be in flight)
    for i in range(3, 6):
        ks[1].execute_device(pos[1][i].data_ptr(), frc[1].data_ptr(), ene[1].data_ptr(), streams[1].cuda_stream)
        ks[0].execute_device(pos[0][i].data_ptr(), frc[0].data_ptr(), ene[0].data_ptr(), streams[0].cuda_stream)
    assert ks[0].finish(streams[0].cuda_stream) == 0 and ks[1].finish(streams[1].cuda_stream) == 0
    oa1, oa2, ob = Oracle(r, g, a, q, h, version=1), Oracle(r, g, a, q2, h, version=1), Oracle(*sb.params(), version=1)
    want_a = [oa1.execute(geoms[0][i]) for i in range(3)] + [oa2.execute(geoms[0][i]) for i in range(3, 6)]
    want_b = [ob.execute(geoms[1][i]) for i in range(6)]
    for w, want in ((0, want_a), (1, want_b)):
        assert abs(ene[w].item() - sum(x[0] for x in want)) < 6 * TIGHT * 10
        assert np.abs(frc[w].cpu().numpy() - sum(x[1] for x in want)).max() < 6 * TIGHT


# ---- round 4: what a timed region of jittered geometries never contains ------------------------------------------------
def _walk(start, steps, sigma, seed):
    rng = np.random.default_rng(seed)
    return start[None] + np.cumsum(rng.normal(0.0, sigma, (steps,) + start.shape), axis=0)


def _predicted_rebuilds(walk, half_skin=0.05):
    """Steps at which k_prep finds an atom further than half the skin from where it was at the last build (the first
    evaluation builds)."""
    ref, out = None, []
    for k, pos in enumerate(walk):
        if ref is None or np.sqrt(((pos - ref) ** 2).sum(axis=1)).max() > half_skin:
            out.append(k)
            ref = pos
    return out


def test_random_walk_on_the_device_path_matches_the_oracle_across_rebuilds(gpu_required, systems, monkeypatch):
    """The drift record of bench.py, checked: evaluations queued through agbnp_hip_execute_device along a cumulative
    random walk (no tethers), the neighbour rows rebuilt on the device whenever an atom has drifted out of half the skin.
    Sampled steps -- the first, every kind of rebuild step, the step after a rebuild, a plain one in between, the last --
    must match the oracle, nothing may be withheld, and the rows must have been rebuilt exactly when predicted."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_ROW_SLICE", "256")  # (a fixed slice length: no extra build while it is being tuned)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    steps = 160
    walk = _walk(s.pos, steps, 0.002, 41)
    rebuilds = _predicted_rebuilds(walk)
    assert len(rebuilds) >= 3 and rebuilds[0] == 0, rebuilds
    sample = sorted({0, 1, rebuilds[1], rebuilds[1] + 1, (rebuilds[1] + rebuilds[2]) // 2, rebuilds[-1], min(rebuilds[-1] + 1, steps - 1), steps - 1})
    dev = torch.device("cuda:0")
    pos = torch.tensor(walk, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((len(sample), s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((len(sample), 1), dtype=torch.float64, device=dev)
    scratch_f = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    scratch_e = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    f0 = np.zeros((s.n, 3))
    k.execute(walk[0], f0)  # (settles the forest packing; builds the rows at walk[0]: the walk's own step 0 does not rebuild)
    b0 = int(k.scalar("row_builds"))
    for step in range(steps):  # ALL queued before one finish, as an MD driver would
        if step in sample:
            i = sample.index(step)
            k.execute_device(pos[step].data_ptr(), frc[i].data_ptr(), ene[i].data_ptr(), stream)
        else:
            k.execute_device(pos[step].data_ptr(), scratch_f.data_ptr(), scratch_e.data_ptr(), stream)
    assert k.finish(stream) == 0, k.withheld()
    assert k.scalar("rows_on") == 1
    assert int(k.scalar("row_builds")) - b0 == len(rebuilds) - 1, (rebuilds, b0, k.scalar("row_builds"))
    got_f, got_e = frc.cpu().numpy(), ene.cpu().numpy()
    for i, step in enumerate(sample):
        eo, fo = oracle.execute(walk[step])
        assert_close(float(got_e[i, 0]), got_f[i], eo, fo)


def test_random_walk_small_system_every_step(gpu_required, systems):
    """trpcage, 300 steps of a faster walk (sigma 0.003 nm), EVERY step against the oracle through the host entry point:
    rebuild steps, the extra slice-tuning rebuild and forest re-plans included."""
    s = systems("trpcage")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    oracle = Oracle(*s.params(), version=1)
    walk = _walk(s.pos, 300, 0.003, 7)
    worst = 0.0
    for pos in walk:
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert_close(e, f, eo, fo)
        worst = max(worst, np.abs(f - fo).max())
    assert int(k.scalar("row_builds")) >= len(_predicted_rebuilds(walk))
    assert int(k.scalar("pack_plans")) >= 300 // 16 - 2  # (a healthy packing is planned anew every sixteenth evaluation, or when the trees have drifted)


def test_every_queued_evaluation_behind_a_truncated_row_is_withheld(gpu_required, systems, monkeypatch):
    """a neighbour row that does not fit its stride used to be flagged only in the evaluation that rebuilt it
    (the stored length was clamped to the stride, so later evaluations saw 'count == cap' and summed a truncated list into
    the caller's buffers).  The length is now stored untruncated: EVERY evaluation queued before the host reacts is
    withheld, the buffers stay untouched, and the repeat (on the tile kernels) completes the sums."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    monkeypatch.setenv("AGBNP_HIP_ROWS", "1")
    monkeypatch.setenv("AGBNP_HIP_ROW_STRIDE", "128")
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    monkeypatch.delenv("AGBNP_HIP_ROW_STRIDE")
    oracle = Oracle(*s.params(), version=1)
    geoms = [s.jittered(k_) for k_ in range(3)]
    want = [oracle.execute(g) for g in geoms]
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(3):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 3  # all three, not just the one that rebuilt the rows
    assert k.withheld() == [0, 1, 2]
    assert float(frc.abs().max()) == 0.0 and float(ene.abs().max()) == 0.0  # nothing of a truncated list reached the caller
    with pytest.raises(P.OpenMMException, match="no completed evaluation"):
        k.vector("born")  # (diagnostics of a void evaluation are not handed out either)
    for attempt in range(4):
        for i in range(3):
            k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
        if k.finish(stream) == 0:
            break
        frc.zero_()
        ene.zero_()
    else:
        raise AssertionError("the repeat did not converge")
    assert k.scalar("rows_on") == 0  # the whole stride was walked already: the tile kernels took over
    assert abs(ene.item() - sum(w[0] for w in want)) < 3 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 3 * TIGHT


@pytest.mark.parametrize("name,cutoff", [("1dwc", 1.0), ("2clr", 1.0), ("1dwc_x4", 1.0)])
def test_fast_single_mode_rows_against_the_cutoff_oracle(gpu_required, systems, name, cutoff):
    """AGBNP_HIP_MODE_FAST | AGBNP_HIP_MODE_SINGLE since round 4: ALL three pair stages compute their pair terms in single
    precision (Born and chain-rule rows too: FP32 table in LDS, positions relative to the group's first row atom), as the
    reference's OpenCL platform does throughout (AGBNPBornRadii.cl:181-430).  Against the FP64 oracle with the same cutoff
    switch, at single-precision tolerances: relative 2e-6 in the energy, 2e-4 of the largest force.  (Parity unpinned: the
    reference holds no vector of its OpenCL platform; the oracle's switch is tied to the pinned path by its limit.)"""
    s = P.lattice(systems("1dwc"), 2, 2, 1, 7.0) if name == "1dwc_x4" else systems(name)
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    force.setCutoffDistance(cutoff)
    k = P.HipCalcAGBNPForceKernel(mode="fast+single")
    k.initialize(force)
    oracle = Oracle(*s.params(), version=1, cutoff=cutoff)
    rng = np.random.default_rng(9)
    for pos in (s.pos, s.pos + rng.normal(0.0, 0.04, s.pos.shape)):  # (the second geometry rebuilds the lists)
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        assert abs(e - eo) < 2e-6 * abs(eo) + 2e-2, (e, eo)
        assert np.abs(f - fo).max() < 2e-4 * np.abs(fo).max(), (np.abs(f - fo).max(), np.abs(fo).max())
        assert np.abs(f - fo).max() > 1e-9  # (and it IS single precision)
    assert k.scalar("rows_on") == 1
    born64 = None
    k64 = P.HipCalcAGBNPForceKernel(mode="fast")
    k64.initialize(force)
    f = np.zeros((s.n, 3))
    k64.execute(s.pos, f)
    born64 = k64.vector("born")
    f = np.zeros((s.n, 3))
    k.execute(s.pos, f)
    assert np.abs(k.vector("born") / born64 - 1.0).max() < 5e-6  # Born radii from FP32 descreening sums
