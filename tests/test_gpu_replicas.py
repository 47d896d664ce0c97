"""GPU box: the N > 1 path of bench.py on real hardware (BASELINE.json configs[4], SURVEY.md s.8e: replicas only -- the
reference itself refuses a multi-device context, platforms/opencl/src/OpenCLAGBNPKernels.cpp:410-411).  The builder's box
has ONE GPU, so two ranks share it (AGBNP_BENCH_BACKEND=gloo: collectives on CPU tensors) and the RCCL plumbing is
rehearsed with one rank (AGBNP_BENCH_FORCE_DIST=1); the driver's 8-GPU node runs the same file with one device per rank.

Every case starts `python3 bench.py ...` as a FRESH child process (subprocess.run): nothing that has initialised HIP is
ever exec'ed, the pytest process only waits.  Three processes use the GPU at most (pytest's own context + two ranks)."""
import json
import os
import re
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags, timeout=240):
    env = dict(os.environ, AGBNP_BENCH_GRACE_SECONDS="20", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AGBNP_BENCH_BACKEND_MODULE"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    pids = [int(m) for m in re.findall(r"bench: rank \d+ of \d+: pid (\d+)", p.stderr)]  # (two ranks' lines may share a line of the pipe)
    return p, lines, pids, time.time() - t0


def _gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    except PermissionError:
        return False
    return False


def test_two_ranks_share_the_gpu_through_the_self_launcher(gpu_required):
    """`python3 bench.py --gpus 2` with no launcher around it, on the real engine: two fresh processes, two contexts on the
    device, one JSON line from rank 0 with the whole-job figure, roofline and cpu_baseline, and EVERY rank's own sample
    against the CPU oracle."""
    p, lines, pids, took = _bench({"AGBNP_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "5", "--warmup", "2", "--secondary", "0",
                                  "--cpu-evals", "1", "--preheat-ms", "50")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 2 and len(r["ranks"]) == 2 and [x["rank"] for x in r["ranks"]] == [0, 1]
    assert len({x["pid"] for x in r["ranks"]}) == 2 and sorted(pids) == sorted(x["pid"] for x in r["ranks"])
    assert r["launcher"].startswith("bench.py self-launch") and r["collectives"] == "gloo"
    assert r["scaling"] == "weak" and r["unit"] == "ns/day" and r["steps"] == 5 and r["warmup"] == 2
    assert abs(r["value"] - 2 * 86.4 / r["ms_per_step"]) < 1e-9 * r["value"]
    assert r["ms_per_step"] >= max(x["ms_per_eval"] for x in r["ranks"]) * (1 - 1e-9)  # the job's time is its slowest rank's
    assert 0.02 < r["ms_per_step"] < 5.0                                              # a real evaluation ran (1dwc: ~0.1-0.2 ms shared)
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["kernel"] == "k_tree_cavity" and 0 < r["roofline"]["frac"] < 1
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["ms_per_eval"] > 50
    for x in r["ranks"]:  # the engine's answer on every rank against the oracle (same bar as the one-GPU parity tests)
        # (forces at the one-GPU tests' bar; the energy, ~ -1e5 kJ/mol, at 1e-6: the same relative bar as tests/test_gpu_parity.py)
        assert x["parity_on_sample"]["max_abs_dE_kJmol"] <= 1e-6 and x["parity_on_sample"]["max_abs_dF_kJmolnm"] <= 1e-7, x
        assert x["device_name"] and x["timed_tries"] >= 1
    assert all(_gone(pid) for pid in pids)
    assert took < 90, took


def test_one_rank_sends_every_collective_through_rccl(gpu_required):
    """The RCCL plumbing on the one GPU there is: a one-rank process group on backend nccl, every Job.sync and the gather
    of the records through it."""
    p, lines, pids, took = _bench({"AGBNP_BENCH_FORCE_DIST": "1"}, "--gpus", "1", "--steps", "5", "--warmup", "2", "--secondary", "0",
                                  "--cpu-evals", "1", "--preheat-ms", "50")
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 1 and r["collectives"] == "nccl" and len(r["ranks"]) == 1
    assert r["ranks"][0]["parity_on_sample"]["max_abs_dF_kJmolnm"] <= 1e-7
    assert "roofline" in r and "cpu_baseline" in r
    assert took < 90, took


def test_a_rank_that_dies_mid_pass_ends_the_job_and_frees_the_gpu(gpu_required):
    """Rank 1 raises between the barriers of its timed pass (after its warm-up and pre-heat have run on the GPU): exit code
    != 0, no JSON line, and neither rank is left behind holding the device."""
    p, lines, pids, took = _bench({"AGBNP_BENCH_BACKEND": "gloo", "AGBNP_BENCH_FAIL_AT": "1:3"},  # (rank 1, third timed evaluation)
                                  "--gpus", "2", "--steps", "5", "--warmup", "2", "--secondary", "0", "--cpu-evals", "0", "--preheat-ms", "50")
    assert p.returncode != 0, p.stdout
    assert lines == []
    assert "injected failure" in p.stderr and "the job ends on every rank" in p.stderr
    assert len(pids) == 2 and all(_gone(pid) for pid in pids), pids
    assert took < 90, took


def _errors(node, path="line"):
    """Every {"error": ...} entry anywhere inside the line, with its path."""
    found = []
    if isinstance(node, dict):
        if "error" in node:
            found.append((path, node["error"]))
        for k, v in node.items():
            found += _errors(v, f"{path}.{k}")
    elif isinstance(node, list):
        for i, v in enumerate(node):
            found += _errors(v, f"{path}[{i}]")
    return found


def test_the_drivers_own_command_line_prints_a_complete_line(gpu_required):
    """The driver's exact command -- `python3 bench.py --gpus 1 --steps 20 --warmup 5`, a fresh child process, every optional
    record behind the headline included (BENCH_r05 died in one of them: no `-m gpu` test ran this command line): exit code 0,
    ONE JSON line with `roofline` and `cpu_baseline`, BASELINE.json's configurations in `secondary`, and no record that
    reports an error or was cut off by the watchdog."""
    p, lines, pids, took = _bench({}, "--gpus", "1", "--steps", "20", "--warmup", "5", timeout=500)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout[-2000:]
    r = lines[0]
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 5 and r["unit"] == "ns/day"
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["avg_launch_us"] > 0
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 1 and r["cpu_baseline"]["ms_per_eval"] > 50
    assert r["parity_on_sample"]["max_abs_dF_kJmolnm"] <= 1e-7
    assert "optional_records_aborted" not in r and "roofline_error" not in r and "cpu_baseline_error" not in r
    assert _errors(r) == [], _errors(r)
    assert [s["workload"] for s in r["secondary"]] == ["trpcage", "trpcage", "1dwc_x4", "2clr"]
    for s in r["secondary"]:
        assert s["parity_on_sample"]["max_abs_dF_kJmolnm"] <= 1e-7, s
    assert {"drift", "other_modes", "concurrent_replicas_on_one_gpu", "openmm_entry", "openmm_entry_particle_order", "md_loop"} <= set(r)
    assert r["openmm_entry"]["launches_per_evaluation"] == 5  # (round 6: the glue's entry point stays in the five-launch mode)
    assert r["secondary"][3]["timed_tries"] <= 2 and r["secondary"][3]["forests"] <= 1280  # 2clr: one round of forests
    assert "did not settle" not in p.stderr
    assert all(_gone(pid) for pid in pids)
    assert took < 240, took
