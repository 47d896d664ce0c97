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


def _fake_replica(bench, torch, script, raise_at=0):
    class _Dev:
        @staticmethod
        def synchronize():
            pass

        @staticmethod
        def wait_idle():
            pass

    rep = bench.Replica.__new__(bench.Replica)  # the real settle / preheat / timed logic, no device
    rep.dev, rep.kernel, rep.stream = _Dev, _FakeKernel(script), None
    rep.d_force, rep.d_energy = torch.zeros(3), torch.zeros(1)
    rep.runs = 0

    def run(first, count):
        rep.runs += 1
        if raise_at and rep.runs == raise_at:
            raise RuntimeError("scripted HIP error")

    rep.run = run
    return rep


def _retry_worker(rank, world, port, out, scripts, raise_at):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rep = _fake_replica(bench, torch, scripts[rank], raise_at.get(rank, 0))
    job = bench.Job(dist, dev, rank)
    try:
        seconds, warm = bench.headline_pass(rep, job, 2, 4, 0.0)
        _, elapsed = job.sync(seconds=seconds)  # the collective that used to be entered by the fast rank alone
        job.sync()
        dist.destroy_process_group()
        out.put((rank, "ok", rep.tries, rep.kernel.calls, elapsed > 0))
    except SystemExit as exc:
        out.put((rank, "exit", int(exc.code), rep.kernel.calls, dist.is_initialized()))


def _run_world(scripts, raise_at=None):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retry_worker, args=(r, 2, port, q, scripts, raise_at or {})) for r in range(2)]
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
    # settle, pre-heat, the first timed try, the untimed recovery pass between two tries (Replica.recover: every rank runs it,
    # the verdict was the job's), the second try: finish() read on every rank every time
    assert results[0][3] == results[1][3] == 5


def test_a_rank_that_never_settles_ends_the_job_on_every_rank():
    """Rank 1's capacity never settles: its failure travels in the next collective, every rank destroys the process group
    and exits 1 -- nobody is left waiting."""
    results = _run_world({0: [], 1: [1] * 64})
    assert [r[1] for r in results] == ["exit", "exit"], results
    assert [r[2] for r in results] == [1, 1], results
    assert [r[4] for r in results] == [False, False], results  # process group destroyed on both


def test_a_rank_that_raises_inside_the_timed_pass_ends_the_job_on_every_rank():
    """rank 1 fails INSIDE its timed pass (a HIP error between the two barriers) while rank 0 sits in the barrier
    behind the timed region.  Every collective of the job is the same operation and carries the failure flag, so rank 1's
    next collective (the end of its phase) meets rank 0's barrier, rank 0 sees the flag there and both leave with exit
    code 1 -- the vote is not mistaken for a 'withheld' count and nobody enters another collective."""
    # run() calls of a rank: 1 = settle, 2 = the timed try (the pre-heat of 0 s runs none)
    results = _run_world({0: [], 1: []}, raise_at={1: 2})
    assert [r[1] for r in results] == ["exit", "exit"], results
    assert [r[2] for r in results] == [1, 1], results
    assert [r[4] for r in results] == [False, False], results


# ---- bench.main end to end: `python3 bench.py --gpus 2` with no launcher around it --------------------------------------------
def _self_launch(extra_env, *flags, timeout=300):
    import json
    import subprocess

    env = dict(os.environ, AGBNP_BENCH_BACKEND="gloo", AGBNP_BENCH_BACKEND_MODULE="tests.fake_bench_backend",
               AGBNP_BENCH_GRACE_SECONDS="20", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--system", "trpcage", "--steps", "6", "--warmup", "2",
                        "--preheat-ms", "0", *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_bench_main_self_launches_two_ranks_end_to_end():
    """The driver's invocation for N > 1 -- `python3 bench.py --gpus N`, no torch.distributed.run -- must work: the parent
    (which imports neither torch nor the engine) starts the ranks, rank 0's ONE JSON line comes back through it, and the
    line carries roofline and cpu_baseline like the one-GPU line.  Rank 1 has one evaluation withheld in its first timed try:
    both ranks repeat."""
    pytest.importorskip("torch")
    p, lines = _self_launch({"AGBNP_FAKE_WITHHELD_1": "0,0,1"}, "--cpu-evals", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 2 and len(r["ranks"]) == 2 and r["distinct_devices"] == 2
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and len({x["pid"] for x in r["ranks"]}) == 2
    assert [x["timed_tries"] for x in r["ranks"]] == [2, 2]
    assert r["launcher"].startswith("bench.py self-launch")
    assert r["scaling"] == "weak" and r["unit"] == "ns/day" and r["steps"] == 6 and r["warmup"] == 2
    assert abs(r["value"] - 2 * 86.4 / r["ms_per_step"]) < 1e-9 * r["value"]
    assert r["ms_per_step"] >= max(x["ms_per_eval"] for x in r["ranks"]) * (1 - 1e-9)  # the job's time is its slowest rank's
    assert r["roofline"]["kernel"] == "k_tree_pseudo" and r["roofline"]["bound"] == "hbm"  # (the fake's slowest kernel)
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["ms_per_eval"] > 0
    assert r["parity_on_sample"]["max_abs_dF_kJmolnm"] == 0.0  # (the fake answers with the oracle itself)
    assert [x["parity_on_sample"]["max_abs_dF_kJmolnm"] for x in r["ranks"]] == [0.0, 0.0]  # every rank checks its own sample
    assert "secondary" not in r


def test_bench_main_a_failing_rank_ends_the_self_launched_job():
    """A rank that dies mid-pass: the parent returns a non-zero exit code, prints no JSON line and leaves no rank behind."""
    pytest.importorskip("torch")
    t0 = __import__("time").time()
    p, lines = _self_launch({"AGBNP_FAKE_RAISE_1": "4"}, "--cpu-evals", "0")
    assert p.returncode == 1, (p.returncode, p.stderr[-2000:])
    assert lines == []
    assert "fake HIP error" in p.stderr
    assert __import__("time").time() - t0 < 120


def test_bench_main_failure_hook_of_the_gpu_tests():
    """AGBNP_BENCH_FAIL_AT=<rank>:<k> (what tests/test_gpu_replicas.py uses on real GPUs): the k-th evaluation of that rank's
    timed pass raises; same outcome as a HIP error."""
    pytest.importorskip("torch")
    p, lines = _self_launch({"AGBNP_BENCH_FAIL_AT": "1:3"}, "--cpu-evals", "0")
    assert p.returncode == 1, (p.returncode, p.stderr[-2000:])
    assert lines == [] and "injected failure at evaluation 3" in p.stderr


def test_the_launching_parent_never_loads_the_gpu_stack():
    """The self-launcher must not touch the GPU: it imports neither torch nor numpy nor the engine."""
    import subprocess
    code = ("import sys; sys.argv = ['bench.py']; import bench; "
            "print([m for m in ('torch', 'numpy', 'openmm_agbnp_plugin_amd') if m in sys.modules])")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "[]", (out.stdout, out.stderr)


def test_bench_main_under_torch_distributed_run():
    """The contract's launch line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- still works: with RANK in the environment the same file is the worker."""
    pytest.importorskip("torch")
    import json
    import subprocess

    env = dict(os.environ, AGBNP_BENCH_BACKEND="gloo", AGBNP_BENCH_BACKEND_MODULE="tests.fake_bench_backend")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--system", "trpcage", "--steps", "6",
                        "--warmup", "2", "--preheat-ms", "0", "--cpu-evals", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 2 and len(r["ranks"]) == 2 and r["distinct_devices"] == 2
    assert r["launcher"] == "torch.distributed.run" and r["collectives"] == "gloo"
    assert "roofline" in r and "cpu_baseline" in r


# ---- the contract's line never depends on an optional record (BENCH_r05: a secondary that would not settle took the line with it) ----
def test_an_optional_record_that_fails_becomes_an_error_entry_and_the_line_is_printed_once(capsys):
    import json

    import bench

    line = bench.Line({"metric": "m", "value": 1.0, "secondary": []})

    def never_settles():
        raise bench.NotSettled("2clr", {"phase": "timed", "tries": 4, "history": [{"withheld": 1, "overflow_kinds": 36, "pack_level": 1, "forests": 2560,
                                                                                  "variant": 0, "withheld_indices": [152], "try_number": 4}]})

    entry = line.optional("secondary: 2clr", never_settles)
    assert entry["error"].startswith("NotSettled: bench: tree capacity did not settle (2clr): timed try 4; last withheld 1 kinds 36 level 1 forests 2560")
    assert entry["report"]["history"][0]["withheld_indices"] == [152]
    line.append("secondary", dict(config="x", **entry))
    assert line.optional("fine", lambda: {"ms_per_eval": 0.1}) == {"ms_per_eval": 0.1}
    line.emit()
    line.emit()  # (the watchdog and the `finally` may both get here: one line)
    out, err = capsys.readouterr()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["secondary"][0]["error"].startswith("NotSettled")
    assert "record 'secondary: 2clr' failed and is reported inside the line" in err


def test_bench_main_prints_the_line_when_the_optional_records_overrun_their_budget():
    """The driver's one-GPU command shape on the CPU double: headline, roofline and cpu_baseline are in the line, the optional
    records behind them get 0.2 s and are cut off by the watchdog -- exit code 0, one line, which says so."""
    pytest.importorskip("torch")
    import json
    import subprocess

    env = dict(os.environ, AGBNP_BENCH_BACKEND="gloo", AGBNP_BENCH_BACKEND_MODULE="tests.fake_bench_backend")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat-ms", "0",
                        "--cpu-evals", "1", "--secondary-budget-s", "0.2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 1 and "roofline" in r and "cpu_baseline" in r and r["config"]["workload"].startswith("1dwc")
    assert r["optional_records_aborted"].startswith("watchdog") and "overran their budget" in p.stderr


def test_bench_main_reports_a_failed_optional_record_inside_the_line():
    """An optional record that dies (AGBNP_BENCH_FAIL_RECORD=drift: the hook raises SystemExit inside `Line.optional`, as the 2clr
    entry of round 5 did on the driver's box): its entry in the line is {"error": ...}, the records behind it still run (until
    the watchdog's budget here), exit code 0, one line with roofline and cpu_baseline."""
    pytest.importorskip("torch")
    import json
    import subprocess

    env = dict(os.environ, AGBNP_BENCH_BACKEND="gloo", AGBNP_BENCH_BACKEND_MODULE="tests.fake_bench_backend", AGBNP_BENCH_FAIL_RECORD="drift")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat-ms", "0",
                        "--cpu-evals", "1", "--secondary-budget-s", "6"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert "roofline" in r and "cpu_baseline" in r
    assert r["drift"]["error"].startswith("SystemExit: injected failure of the record 'drift'")
    assert "record 'drift' failed and is reported inside the line" in p.stderr
    assert "secondary" in r  # (the records behind the failed one were started)
