"""CPU, world_size 2, gloo: the only cross-rank traffic of the replica benchmark (MAX of elapsed time and the
all-gather of per-replica throughput records; bench.py s. 'multi-GPU') behaves as on RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cpu")
    elapsed = bench.max_over_ranks(dist, 1.0 + rank, dev)          # slowest rank defines the job time
    recs = bench.gather_throughput(dist, 100.0 + rank, 0.5 + rank, dev)
    ident = bench.gather_records(dist, {"rank": rank, "device_index": rank, "ms_per_eval": 0.5 + rank}, dev)
    world = dist.get_world_size()  # what bench.py reports as n_gpus
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, elapsed, recs, ident, world))


def test_world_size_two_gather_and_max():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, recs, ident, world in results:
        assert elapsed == 2.0
        assert recs == [(100.0, 0.5), (101.0, 1.5)]
        assert world == 2
        assert [r["rank"] for r in ident] == [0, 1] and [r["device_index"] for r in ident] == [0, 1]


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    torch = pytest.importorskip("torch")
    import bench
    assert bench.max_over_ranks(None, 0.25, torch.device("cpu")) == 0.25
    assert bench.gather_throughput(None, 146.0, 0.59, torch.device("cpu")) == [(146.0, 0.59)]
    assert bench.gather_records(None, {"rank": 0}, torch.device("cpu")) == [{"rank": 0}]
    total, per_kernel = bench.algorithmic_bytes(4152, 216146)
    assert total == 216146 * 128 * 8 + 2145 * 4096 * 4 + 4152 * 104  # 257 MB for 1dwc
    assert abs(total - 257e6) < 1e6
    assert sum(per_kernel.values()) == total


# ---- the retry / abort decisions of the multi-rank bench are collective ------------------------------------------------
class _FakeKernel:
    """Stands in for HipCalcAGBNPForceKernel: finish() plays back a script of withheld counts (then 0 for ever)."""

    def __init__(self, script):
        self.script = list(script)
        self.calls = 0

    def finish(self, stream):
        self.calls += 1
        return self.script.pop(0) if self.script else 0


def _fake_replica(bench, torch, script):
    class _Cuda:
        @staticmethod
        def synchronize():
            pass

    class _Torch:
        cuda = _Cuda

    rep = bench.Replica.__new__(bench.Replica)  # the real settle / preheat / timed logic, no device
    rep.torch, rep.kernel, rep.stream = _Torch, _FakeKernel(script), None
    rep.d_force, rep.d_energy = torch.zeros(3), torch.zeros(1)
    rep.run = lambda first, count: None
    return rep


def _retry_worker(rank, world, port, out, scripts):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rep = _fake_replica(bench, torch, scripts[rank])
    phases = bench.Phases(dist, dev, rank)
    try:
        seconds, warm = bench.headline_pass(rep, dist, dev, 2, 4, 0.0, phases)
        elapsed = bench.max_over_ranks(dist, seconds, dev)  # the collective that used to be entered by the fast rank alone
        dist.barrier()
        dist.destroy_process_group()
        out.put((rank, "ok", rep.tries, rep.kernel.calls, elapsed > 0))
    except SystemExit as exc:
        out.put((rank, "exit", int(exc.code), rep.kernel.calls, dist.is_initialized()))


def _run_world(scripts):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retry_worker, args=(r, 2, port, q, scripts)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_one_rank_withheld_makes_every_rank_repeat():
    """Rank 1 reports one withheld evaluation in its first timed try (after a clean warm-up and pre-heat); rank 0 never does.
    Both ranks must make the same number of tries and reach the closing collectives together (the run used to hang here:
    rank 0 went on to max_over_ranks while rank 1 was back in the barrier of its second try)."""
    results = _run_world({0: [], 1: [0, 0, 1]})
    assert [r[1] for r in results] == ["ok", "ok"], results
    assert [r[2] for r in results] == [2, 2], results  # equal try counts
    assert results[0][3] == results[1][3] == 4         # settle, pre-heat, two timed tries: finish() read on every rank every time


def test_a_rank_that_never_settles_ends_the_job_on_every_rank():
    """Rank 1's capacity never settles: its failure is all-reduced, every rank destroys the process group and exits 1 --
    nobody is left waiting."""
    results = _run_world({0: [], 1: [1] * 64})
    assert [r[1] for r in results] == ["exit", "exit"], results
    assert [r[2] for r in results] == [1, 1], results
    assert [r[4] for r in results] == [False, False], results  # process group destroyed on both
