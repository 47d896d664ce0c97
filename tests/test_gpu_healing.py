"""GPU box: forests that outgrow their store are HEALED inside the tree launch (round 6; csrc/tree_kernels.hip,
cavity_forests).  The reference's tree simply grows (gaussvol/gaussvol.cpp:154-192: `overlaps.insert` at the tail of a
vector); the engine's stores are fixed, and its forest packing is planned from an EARLIER geometry's subtree shapes.  Rounds
2-5 voided the whole evaluation when a forest did not fit (withheld, repeated by the host unpacked, the packing tightened for
everybody -- BENCH_r05 died of exactly that on 2clr).  Now the workgroup builds the forest's work items again in smaller sets
(the later sets into spare work slots that the pseudo-volume launch replays through its queue), and a lone item that outgrows
the store is built as the parts of a four-way share: the evaluation is COMPLETE, the numbers are the oracle's."""
import numpy as np
import pytest

import openmm_agbnp_plugin_amd as P
from oracle import Oracle

pytestmark = pytest.mark.gpu
TIGHT = 1e-7


def _queue(k, torch, geoms, n):
    dev = torch.device("cuda:0")
    pos = torch.tensor(np.stack(geoms), dtype=torch.float64, device=dev).contiguous()
    frc = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    ene = torch.zeros((1,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda i: k.execute_device(pos[i].data_ptr(), frc.data_ptr(), ene.data_ptr(), stream)
    return run, frc, ene, stream


def _freeze_eight_subtrees_per_forest(k, nheavy):
    """A packing frozen from the host (test hook agbnp_debug_set_packing): forest f = the whole subtrees of heavy atoms
    8 f .. 8 f + 7, whatever their size."""
    import ctypes as C

    from openmm_agbnp_plugin_amd import _lib
    lib = _lib.load()
    lib.agbnp_debug_set_packing.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    nf = (nheavy + 7) // 8
    order = (C.c_int * nheavy)(*range(nheavy))
    start = (C.c_int * (nf + 1))(*[min(8 * f, nheavy) for f in range(nf + 1)])
    assert lib.agbnp_debug_set_packing(k._h, order, nheavy, start, nf, 1) == _lib.OK
    return nf


@pytest.mark.parametrize("launches,version", [("six", 1), ("five", 1), ("five", 0)])
def test_overfull_forests_are_healed_inside_the_tree_launch(gpu_required, systems, monkeypatch, launches, version):
    """1dwc with a packing frozen at EIGHT whole subtrees per forest (261 forests of ~830 nodes for stores of 432): not one of
    them fits.  Every forest is built again in halves (and halves of halves), the later sets in spare work slots; nothing is
    withheld, energy and forces -- the pseudo-volume replay of the spare slots included -- are the oracle's on every one of a
    queue of jittered geometries, scalar 17 counts the healed sets, no capacity variant is raised.  Both launch chains: the
    five-launch mode's forest workgroups read the caller's positions themselves, also for the sets they build again; and
    version 0 (no pseudo-volume replay: the spare slots only carry shapes and energies)."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("AGBNP_HIP_FIVE_LAUNCHES", "0" if launches == "six" else "1")
    s = systems("1dwc")
    oracle = Oracle(*s.params(), version=version)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=version))
    geoms = [s.jittered(step) for step in range(5)]
    want = [oracle.execute(g) for g in geoms]
    run, frc, ene, stream = _queue(k, torch, geoms, s.n)
    for i in (0, 1):  # settle on the engine's own packing
        run(i)
    assert k.finish(stream) == 0
    assert int(k.scalar("launches")) == ((5 if launches == "five" else 6) if version == 1 else 2)
    nf = _freeze_eight_subtrees_per_forest(k, s.nheavy)
    frc.zero_()
    ene.zero_()
    gen = k.generation()
    for i in range(5):
        run(i)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("healed_forests")) >= 5 * nf  # (every forest of every evaluation, most of them more than once)
    assert abs(ene.item() - sum(w[0] for w in want)) < 5 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 5 * TIGHT
    assert k.generation() == gen and int(k.scalar("variant")) == 0 and int(k.scalar("pack_level")) <= 1
    assert int(k.scalar("total_nodes")) > 200000  # (the subtree shapes of a healed evaluation are complete)


def test_a_planned_packing_that_the_next_geometry_outgrows_is_healed_and_planned_anew(gpu_required, systems, monkeypatch):
    """The real protocol on the six-launch chain (the geometry jumps): the packing is planned on a swollen 1dwc (largest subtree
    58 nodes) and told not to spread the forests over the idle workgroups (five to eight subtrees per forest); the next geometry
    is the real molecule (216 k nodes): the forests do not fit, every single subtree does.  Rounds 2-5 withheld that evaluation
    (tests/test_gpu_parity.py::test_queued_evaluations_withhold_an_overflowed_one keeps that protocol under test with
    AGBNP_HIP_HEAL=0); now nothing is withheld, and the evaluation after the healed one has planned anew from ITS shapes."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("AGBNP_HIP_FIVE_LAUNCHES", "0")
    monkeypatch.setenv("AGBNP_HIP_ROUND_PERMILLE", "100")
    s = systems("1dwc")
    centre = s.pos.mean(axis=0)
    scaled = lambda pos, f: centre + f * (pos - centre)
    oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel()
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    geoms = [scaled(s.jittered(0), 1.3), scaled(s.jittered(1), 1.3), s.pos, s.jittered(2), s.jittered(3)]
    want = [oracle.execute(g) for g in geoms]
    run, frc, ene, stream = _queue(k, torch, geoms, s.n)
    for i in (0, 1):  # settle: the second evaluation already runs on packed forests
        run(i)
    assert k.finish(stream) == 0
    assert int(k.scalar("forests")) <= 400
    frc.zero_()
    ene.zero_()
    gen = k.generation()
    for i in range(5):
        run(i)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("healed_forests")) > 100  # (every forest of the third evaluation, most of them more than once)
    assert abs(ene.item() - sum(w[0] for w in want)) < 5 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 5 * TIGHT
    assert k.generation() == gen and int(k.scalar("variant")) == 0  # no capacity variant was raised for it
    frc.zero_()
    ene.zero_()
    for i in (3, 4):
        run(i)
    assert k.finish(stream) == 0 and int(k.scalar("healed_forests")) == 0
    assert abs(ene.item() - (want[3][0] + want[4][0])) < 2 * TIGHT


def test_a_lone_subtree_beyond_the_store_is_built_in_four_parts_at_once(gpu_required, systems):
    """A FRESH context on 2clr, every evaluation queued before anybody reads the log: the first one runs one whole subtree per
    work slot and meets subtrees of up to 479 nodes in a 432-node store.  Rounds 4-5 withheld that evaluation (and the device
    shared the subtree from the next one on); now the workgroup builds the item again as the four parts of a four-way share,
    three of them into spare slots: NO evaluation is withheld, all six sums are the oracle's, the engine stays on the smallest
    store and the packing's level at 0."""
    torch = pytest.importorskip("torch")
    s = systems("2clr")
    oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    geoms = [s.jittered(40 + step) for step in range(6)]
    run, frc, ene, stream = _queue(k, torch, geoms, s.n)
    for i in range(6):
        run(i)
    assert k.finish(stream) == 0, (list(k.withheld()), int(k.scalar("overflow_kinds")))
    assert int(k.scalar("healed_forests")) >= 1  # (one spare set per item refined from a two-way share, three from a whole subtree)
    assert int(k.scalar("variant")) == 0 and int(k.scalar("pack_level")) == 0
    assert int(k.scalar("max_subtree_nodes")) > 432
    want = [oracle.execute(g) for g in geoms]
    assert abs(ene.item() - sum(w[0] for w in want)) < 6 * TIGHT
    assert np.abs(frc.cpu().numpy() - sum(w[1] for w in want)).max() < 6 * TIGHT
    assert int(k.scalar("forests")) <= 1280  # (and the rounds rule has packed them into one round by now)


def test_a_mid_size_system_never_withholds_a_jittered_evaluation(gpu_required, systems):
    """VERDICT r05 item 2.  2clr under the rounds rule packs its forests to 0.88 of the fill target; four of the 200 jittered
    geometries of bench.py's secondary entry (seeds 7000 + 20 + {21, 146, 152, 161}) outgrow a forest under most plans
    (profiles/r06/probe_2clr_r05_protocol.log: 8 of 36 protocol runs of round 5 did not settle, 45 of their 51 withheld
    evaluations were those four).  600 evaluations in three chunks over those four and four ordinary ones, every chunk read
    with finish(): none withheld in ANY chunk (the first included: a fresh context), the packing stays at one round and level
    0, and the sums are the oracle's."""
    torch = pytest.importorskip("torch")
    s = systems("2clr")
    oracle = Oracle(*s.params(), version=1)
    k = P.HipCalcAGBNPForceKernel(device=0)
    k.initialize(P.AGBNPForce.from_arrays(*s.params(), version=1))
    seeds = [7020 + i for i in (21, 146, 152, 161, 3, 77, 100, 190)]
    geoms = [s.jittered(seed) for seed in seeds]
    want = [oracle.execute(g) for g in geoms]
    run, frc, ene, stream = _queue(k, torch, geoms, s.n)
    rng = np.random.default_rng(6)
    healed = 0
    for chunk in range(3):
        order = rng.integers(0, len(geoms), size=200)
        frc.zero_()
        ene.zero_()
        for i in order:
            run(int(i))
        assert k.finish(stream) == 0, (chunk, list(k.withheld())[:8], int(k.scalar("overflow_kinds")))
        healed += int(k.scalar("healed_forests"))
        counts = np.bincount(order, minlength=len(geoms))
        e_want = sum(c * w[0] for c, w in zip(counts, want))
        f_want = sum(c * w[1] for c, w in zip(counts, want))
        assert abs(ene.item() - e_want) < 200 * TIGHT * max(1.0, abs(want[0][0]) * 1e-3)
        assert np.abs(frc.cpu().numpy() - f_want).max() < 200 * TIGHT
        assert int(k.scalar("forests")) <= 1280 and int(k.scalar("pack_level")) == 0 and int(k.scalar("variant")) == 0
    assert healed > 0  # (the four geometries did what they are here for: without healing this run withholds evaluations)
