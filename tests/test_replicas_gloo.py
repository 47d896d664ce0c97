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
