"""GPU box: the five-launch mode (the default for version 1; the fixture below pins it): no k_prep launch -- the cavity
launch's trailing workgroups do its per-atom work, the tree accumulators / subtree shapes / per-evaluation status words
alternate between two sets, the tree reads the caller's positions itself, the level-2 neighbour masks carry a skin and are
laid down anew ON THE DEVICE when a heavy atom has used up a quarter of it.  Same numbers as the oracle, same contracts as the
default mode (reference path: ReferenceAGBNPKernels.cpp:274-795)."""
import os

import numpy as np
import pytest

import openmm_agbnp_plugin_amd as P
from oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = 1e-7


@pytest.fixture()
def five(monkeypatch):
    monkeypatch.setenv("AGBNP_HIP_FIVE_LAUNCHES", "1")


def _kernel(s, version=1):
    k = P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=version))
    return k


def _close(e, f, eo, fo, tol=TIGHT):
    assert abs(e - eo) < tol * max(1.0, abs(eo) * 1e-3), f"energy differs by {abs(e - eo):.3e}"
    assert np.abs(f - fo).max() < tol, f"forces differ by {np.abs(f - fo).max():.3e}"


@pytest.mark.parametrize("name", ["trpcage", "1dwc", "2clr", "fixture264"])
def test_five_launches_match_the_oracle(gpu_required, systems, five, name):
    """Unrelated geometries one after the other through the host entry point: every jump beyond the masks' skin voids an
    evaluation, the device lays the masks down anew, the call repeats itself -- the caller sees the oracle's numbers."""
    s = systems(name)
    k = _kernel(s)
    oracle = Oracle(*s.params(), version=1)
    centre = s.pos.mean(axis=0)
    for pos in (s.pos, s.jittered(1), s.jittered(2, sigma=0.02), centre + 0.97 * (s.pos - centre), s.pos):
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        _close(e, f, eo, fo)
    times = k.kernel_times() if hasattr(k, "kernel_times") else {}
    assert "k_prep" not in {n for n, v in times.items() if v[1] > 0}


def test_five_launches_is_five_launches(gpu_required, systems, five):
    """The per-kernel timeline of a settled 1dwc evaluation holds five kernels, none of them k_prep."""
    s = systems("1dwc")
    k = _kernel(s)
    f = np.zeros((s.n, 3))
    for step in range(4):
        k.execute(s.jittered(step), f)
    k.set_profiling(True)
    for step in range(4, 8):
        k.execute(s.jittered(step), f)
    times = {n: v for n, v in k.kernel_times().items() if v[1] > 0}
    k.set_profiling(False)
    assert set(times) == {"k_tree_cavity", "k_born_rows", "k_gb_tiles", "k_dborn_rows", "k_tree_pseudo"}, times


def test_masks_heal_on_the_device_along_a_queued_walk(gpu_required, systems, five):
    """A cumulative random walk queued on the device-resident path (nobody reads the log in between): atoms use up the masks'
    skin every few dozen steps, the masks are renewed by the evaluation that notices (a quarter of the skin: still exact),
    no evaluation is withheld, and the sums are the oracle's."""
    torch = pytest.importorskip("torch")
    s = systems("trpcage")
    k = _kernel(s)
    oracle = Oracle(*s.params(), version=1)
    rng = np.random.default_rng(7)
    steps = 150
    walk = s.pos + np.cumsum(rng.normal(0.0, 0.0015, (steps,) + s.pos.shape), axis=0)
    dev = torch.device("cuda:0")
    pos = torch.tensor(walk, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    f0 = np.zeros((s.n, 3))
    k.execute(s.pos, f0)  # (settles the capacity variant and the first masks)
    for i in range(steps):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0, (k.withheld(), int(k.scalar("overflow_kinds")))
    want = [oracle.execute(g) for g in walk]
    assert abs(ene.item() - sum(w[0] for w in want)) < steps * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < steps * TIGHT
    assert float(np.linalg.norm(walk[-1] - s.pos, axis=1).max()) > 0.03  # (the walk did leave the first masks' skin)


def test_a_jump_voids_one_queued_evaluation_only(gpu_required, systems, five):
    """A geometry that jumps (every atom 0.05 nm away at once) in the middle of a queue: that evaluation is withheld (its trees
    were built from masks that no longer cover it), the device renews the masks in the same evaluation, the ones queued
    behind it are complete."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    k = _kernel(s)
    oracle = Oracle(*s.params(), version=1)
    f0 = np.zeros((s.n, 3))
    k.execute(s.pos, f0)
    k.execute(s.jittered(1), f0)
    shifted = s.pos + np.random.default_rng(3).normal(0.0, 0.03, s.pos.shape)
    geoms = [s.jittered(2), s.jittered(3), shifted, shifted + 0.001, s.jittered(4, sigma=0.001) + (shifted - s.pos)]
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(len(geoms)):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 1 and list(k.withheld()) == [2]
    assert int(k.scalar("overflow_kinds")) & 16  # (reported like a reordered context: the evaluation's inputs were stale)
    want = [oracle.execute(g) for g in geoms]
    clean = [0, 1, 3, 4]
    assert abs(ene.item() - sum(want[i][0] for i in clean)) < 4 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(want[i][1] for i in clean)).max() < 4 * TIGHT


def test_a_captured_evaluation_replays_in_the_mode(gpu_required, systems, five):
    """A captured graph replays the same kernel arguments every time; which of the two sets of accumulators an evaluation
    works on is decided on the device (the parity of a counter that the bookkeeping role advances), so a ONE-evaluation graph
    alternates like eager launches do -- also with eager evaluations between its replays."""
    torch = pytest.importorskip("torch")
    s = systems("trpcage")
    k = _kernel(s)
    oracle = Oracle(*s.params(), version=1)
    f0 = np.zeros((s.n, 3))
    for step in range(3):
        k.execute(s.jittered(step), f0)
    dev = torch.device("cuda:0")
    pos = torch.tensor(s.pos, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        frc.zero_()
        ene.zero_()
        k.execute_device(pos.data_ptr(), frc.data_ptr(), ene.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for step in (5, 6, 7, 8, 9):
        geom = s.jittered(step)
        pos.copy_(torch.tensor(geom, dtype=torch.float64))
        g.replay()
        torch.cuda.synchronize()
        eo, fo = oracle.execute(geom)
        assert abs(ene.item() - eo) < TIGHT and np.abs(frc.cpu().numpy() - fo).max() < TIGHT
        if step in (6, 8):  # an eager evaluation between two replays (an odd number of them: the parities still alternate)
            fe = np.zeros((s.n, 3))
            ee = k.execute(s.jittered(20 + step), fe)
            eo2, fo2 = oracle.execute(s.jittered(20 + step))
            _close(ee, fe, eo2, fo2)
    assert k.finish(torch.cuda.current_stream().cuda_stream) == 0
    assert int(k.scalar("launches")) == 5


@pytest.mark.parametrize("precision", ["double", "mixed"])
def test_the_openmm_entry_point_runs_in_the_mode(gpu_required, systems, five, precision):
    """Round 6: agbnp_hip_execute_openmm -- the entry the OpenMM glue calls (openmm_glue/HipAGBNPKernels.cpp, mirror of
    OpenCLAGBNPKernels.cpp:3510-4216) -- stays in the five-launch mode: the forest workgroups read the context's posq (double4, or
    float4 + correction) at the context's slots, the trailing workgroups check the atom order.  A queue of jittered geometries in
    a shuffled, padded context order: five launches, no k_prep, the oracle's sums in the fixed-point planes; then the SAME
    context through agbnp_hip_execute_device and back (the words beside the work-slot rows are rewritten for the other entry
    point), then a reorder of the context's atoms (that evaluation alone is withheld, the repeat is right)."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    n, padded = s.n, (s.n + 31) // 32 * 32
    rng = np.random.default_rng(11)
    oracle = Oracle(*s.params(), version=1)
    k = _kernel(s)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    index = torch.zeros(padded, dtype=torch.int32, device=dev)
    fixed = torch.zeros(3 * padded, dtype=torch.int64, device=dev)
    ebuf = torch.zeros(8, dtype=torch.float64, device=dev)
    state = {}

    def set_order():
        order = rng.permutation(n).astype(np.int32)
        index.copy_(torch.tensor(np.concatenate([order, np.arange(n, padded, dtype=np.int32)])))  # same address, new contents
        state["order"] = order

    def context_arrays(pos):  # -> posq (+ correction) tensors in the context's order, and the positions the engine really sees
        host = np.zeros((padded, 4))
        host[:n, :3] = pos[state["order"]]
        if precision == "double":
            return torch.tensor(host, dtype=torch.float64, device=dev), None, pos
        hi = host.astype(np.float32)
        lo = (host - hi.astype(np.float64)).astype(np.float32)
        seen = np.zeros((n, 3))
        seen[state["order"]] = (hi.astype(np.float64) + lo.astype(np.float64))[:n, :3]
        return torch.tensor(hi, device=dev), torch.tensor(lo, device=dev), seen

    def run_openmm(geoms):
        keep, want_e, want_f = [], 0.0, np.zeros((n, 3))
        for g in geoms:
            posq, corr, seen = context_arrays(g)
            keep.append((posq, corr))
            k.execute_openmm(posq.data_ptr(), precision == "double", corr.data_ptr() if corr is not None else 0, index.data_ptr(), padded,
                             fixed.data_ptr(), ebuf.data_ptr(), True, 3, stream)
            eo, fo = oracle.execute(seen)
            want_e, want_f = want_e + eo, want_f + fo
        return keep, want_e, want_f

    def check(want_e, want_f, evaluations):
        torch.cuda.synchronize()
        got = fixed.cpu().numpy().reshape(3, padded).astype(np.float64) / 2.0 ** 32
        assert not got[:, n:].any()
        assert np.abs(got[:, :n].T - want_f[state["order"]]).max() < evaluations * 1e-6  # (the fixed point resolves 2^-32 per add)
        assert abs(ebuf.cpu().numpy()[3] - want_e) < evaluations * TIGHT * max(1.0, abs(want_e) * 1e-3 / evaluations)
        fixed.zero_()
        ebuf.zero_()

    set_order()
    torch.cuda.synchronize()
    keep, we, wf = run_openmm([s.jittered(199)])  # (a fresh context lays its first neighbour masks down with a launch of its own)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    check(we, wf, 1)
    k.set_profiling(True)
    keep, we, wf = run_openmm([s.jittered(200 + i) for i in range(5)])
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("launches")) == 5
    times = {name for name, v in k.kernel_times().items() if v[1] > 0}
    assert "k_prep" not in times and len(times) == 5, times
    k.set_profiling(False)
    check(we, wf, 5)
    # the same context through the device-resident entry point (positions in particle order), and back
    pos = torch.tensor(np.stack([s.jittered(205), s.jittered(206)]), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    for i in range(2):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0
    w = [oracle.execute(s.jittered(205 + i)) for i in range(2)]
    _close(ene.item(), frc.cpu().numpy(), w[0][0] + w[1][0], w[0][1] + w[1][1], tol=2 * TIGHT)
    keep, we, wf = run_openmm([s.jittered(207), s.jittered(208)])
    assert k.finish(stream) == 0 and int(k.scalar("launches")) == 5
    check(we, wf, 2)
    # OpenMM's reorderAtoms(): same arrays, new contents
    set_order()
    torch.cuda.synchronize()
    keep, we, wf = run_openmm([s.jittered(209)])
    assert k.finish(stream) == 1 and list(k.withheld()) == [0]
    torch.cuda.synchronize()
    assert not fixed.cpu().numpy().any() and not ebuf.cpu().numpy().any()  # nothing of the stale evaluation arrived
    keep, we, wf = run_openmm([s.jittered(209), s.jittered(210)])
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("launches")) == 5
    check(we, wf, 2)


@pytest.mark.parametrize("name", ["trpcage", "1dwc", "fixture264"])
def test_version_0_runs_in_two_launches(gpu_required, systems, five, name):
    """Round 6: GaussVol / GVolSA (version 0, BASELINE.json configs[0]) without its k_prep launch -- the cavity launch with its
    trailing workgroups, then the output launch (energy + bookkeeping roles, forces, the masks' renewal at its tail).  Unrelated
    geometries through the host entry point (jumps: void, renewed, repeated), then a queue of small steps on the device path
    that walks out of the first masks' skin with nothing withheld; the oracle's numbers (ReferenceAGBNPKernels.cpp:152-271)."""
    torch = pytest.importorskip("torch")
    s = systems(name)
    k = _kernel(s, version=0)
    oracle = Oracle(*s.params(), version=0)
    centre = s.pos.mean(axis=0)
    for pos in (s.pos, s.jittered(1), s.jittered(2, sigma=0.02), centre + 0.97 * (s.pos - centre), s.pos):
        f = np.zeros((s.n, 3))
        e = k.execute(pos, f)
        eo, fo = oracle.execute(pos)
        _close(e, f, eo, fo)
    assert int(k.scalar("launches")) == 2
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    steps = 60
    walk = s.pos + np.cumsum(rng.normal(0.0, 0.003, (steps,) + s.pos.shape), axis=0)  # ends ~0.04 nm per atom from where it began
    pos = torch.tensor(walk, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    k.set_profiling(True)
    for i in range(steps):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    times = {n for n, v in k.kernel_times().items() if v[1] > 0}
    assert times == {"k_tree_cavity", "k_outputs"}, times
    k.set_profiling(False)
    sample = [0, 17, 38, steps - 1]
    frc.zero_()
    ene.zero_()
    for i in sample:  # (jumps between the samples: through the host entry point, which repeats a voided evaluation by itself)
        f = np.zeros((s.n, 3))
        e = k.execute(walk[i], f)
        eo, fo = oracle.execute(walk[i])
        _close(e, f, eo, fo)


@pytest.mark.parametrize("flavour", ["deterministic", "tiles", "fast+single"])
def test_the_other_pair_stage_forms_stay_in_the_mode(gpu_required, systems, five, monkeypatch, flavour):
    """Round 6: pair stages other than the FP64 row form keep the five-launch chain -- the tile kernels (the deterministic mode,
    AGBNP_HIP_ROWS=0: the masks' renewal rides at the tail of the GB tile launch) and the single-precision rows of the fast mode
    (at the tail of its Born rows).  A queued walk of small steps that leaves the first masks' skin: five launches, nothing
    withheld, and the mode's numbers -- the oracle's (tiles, deterministic; ReferenceAGBNPKernels.cpp:274-795), or the cutoff
    oracle's at single-precision tolerances (fast+single; parity unpinned: DESIGN.md s.3)."""
    torch = pytest.importorskip("torch")
    s = systems("1dwc")
    force = P.AGBNPForce.from_arrays(*s.params(), version=1)
    if flavour == "tiles":
        monkeypatch.setenv("AGBNP_HIP_ROWS", "0")
    if flavour == "fast+single":
        force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
        force.setCutoffDistance(1.0)
        oracle = Oracle(*s.params(), version=1, cutoff=1.0)
    else:
        oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel(device=0, mode=flavour) if flavour != "tiles" else P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(force)
    rng = np.random.default_rng(4)
    steps = 50
    walk = s.pos + np.cumsum(rng.normal(0.0, 0.003, (steps,) + s.pos.shape), axis=0)
    dev = torch.device("cuda:0")
    pos = torch.tensor(walk, dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((s.n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    k.execute_device(pos[0].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)  # (a fresh context's first masks: a launch of their own)
    assert k.finish(stream) == 0
    k.set_profiling(True)
    for i in range(1, steps):
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("launches")) == 5
    times = {n for n, v in k.kernel_times().items() if v[1] > 0}
    assert "k_prep" not in times and len(times) == 5, times
    k.set_profiling(False)
    frc.zero_()
    ene.zero_()
    for i in (steps - 3, steps - 2, steps - 1):  # (consecutive steps of the walk: no jump)
        k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    assert k.finish(stream) == 0
    want = [oracle.execute(walk[i]) for i in (steps - 3, steps - 2, steps - 1)]
    e_want, f_want = sum(w[0] for w in want), sum(w[1] for w in want)
    if flavour == "fast+single":
        assert abs(ene.item() - e_want) < 2e-6 * abs(e_want) + 6e-2
        assert np.abs(frc.cpu().numpy() - f_want).max() < 6e-4 * np.abs(f_want).max() / 3
    else:
        _close(ene.item(), frc.cpu().numpy(), e_want, f_want, tol=3 * TIGHT)
