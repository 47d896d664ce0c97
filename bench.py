#!/usr/bin/env python3
"""Benchmark of the one hot path: AGBNP1 energy+force evaluation of thrombin (1dwc, 4152 atoms) on MI355X.

A "step" = one complete AGBNP1 evaluation (positions resident in HBM -> forces and energy resident in
HBM) of one replica on a new jittered geometry (file coordinates + N(0, 0.002 nm), so
the overlap tree is rebuilt from a different geometry every step, as in MD).

metric / value : aggregate AGBNP-force-limited ns/day over all replicas at the 1 fs step of the
                 reference's example/1dwc_benchmark.py:20  ( ns/day = 86.4 / ms_per_eval per replica ).
multi-GPU      : replicas only (the force evaluation does not shard).  One process per
                 GPU, independent geometries, no data-path collective; RCCL carries only the job's
                 decisions (MAX of {failed, withheld, elapsed}) and the gather of the per-rank records.
secondary      : on one GPU the same JSON line also carries BASELINE.json's other configurations (trpcage
                 GaussVol, trpcage AGBNP1 with CutoffNonPeriodic 1.2 nm, the 16 608-atom HIV-RT stand-in),
                 each with its own parity-on-sample figure, under "other_modes" the timings of the fast
                 (OpenCL-semantics), fast+single and deterministic modes on the headline workload, and what the
                 headline's protocol never contains: a rebuild evaluation of the neighbour rows and a
                 2000-step random walk ("drift").  --secondary 0 skips them.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N --steps K --warmup W          # no launcher: this process starts the N ranks itself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Only the standard library is imported at module level: the self-launching parent of `--gpus N` must never touch the
GPU (it starts N fresh python processes; nothing that has initialised HIP is ever re-exec'ed), so numpy, torch and the
engine load on first use, in the worker.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Lazy:
    """A module that is imported when somebody first asks it for something."""

    def __init__(self, name):
        self.__dict__["_name"], self.__dict__["_mod"] = name, None

    def __getattr__(self, attr):
        if self.__dict__["_mod"] is None:
            self.__dict__["_mod"] = importlib.import_module(self.__dict__["_name"])
        return getattr(self.__dict__["_mod"], attr)


np = _Lazy("numpy")
P = _Lazy("openmm_agbnp_plugin_amd")  # (loads the HIP runtime shared with torch)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SIMDS = 1024           # 256 CUs x 4
CLOCK_GHZ = 2.4        # max clock; the chip holds ~2.3 GHz in these kernels (profiles/r02/wg_timeline.txt)

# Fallback only (no counter summary under profiles/): vector instructions of one pair step of one wave (64 pairs), read
# off the gfx950 assembly of the shipped kernels: total VALU and the FP64 ones among them.  {kernel: {tile kind: (valu, f64)}}
# tests/test_host_api.py holds this table to within 3 % of the counters of the newest profiles/rNN/pmc_utilization.csv.
PAIR_STEP_VALU = {
    "k_gb_tiles": {"all": (59, 44)},
    "k_born_tiles": {"hh": (49.6, 35), "hl": (35.9, 26)},   # (loop body + the tile's prologue / epilogue spread over its steps)
    "k_dborn_tiles": {"hh": (75, 56), "hl": (52.7, 40)},
}
ISSUE_BOUND_KERNELS = ("k_gb_tiles", "k_born_tiles", "k_dborn_tiles", "k_born_rows", "k_dborn_rows", "k_gb_rows")


def newest_profile_dir():
    """The newest profiles/rNN that holds a counter pass (pmc_utilization.csv): a round's directory exists from its first log on,
    the counters arrive with scripts/profile_round.sh."""
    root = os.path.join(ROOT, "profiles")
    tags = sorted(d for d in os.listdir(root) if d.startswith("r") and d[1:].isdigit() and os.path.isdir(os.path.join(root, d))) if os.path.isdir(root) else []
    full = [t for t in tags if os.path.exists(os.path.join(root, t, "pmc_utilization.csv"))]
    return os.path.join(root, (full or tags)[-1]) if tags else None


def profile_head(directory=None):
    """What the committed counter summaries were measured on (profiles/rNN/profile_head.json, written by
    scripts/profile_round.sh: the library's build id = SHA-256 of its sources, the git head, the date) and whether that is
    the library this process is running.  None when the newest profile carries no such record (rounds 1-4)."""
    d = directory or newest_profile_dir()
    path = os.path.join(d, "profile_head.json") if d else None
    if not path or not os.path.exists(path):
        return {"profile": os.path.relpath(d, ROOT) if d else None, "library_build_id": None, "git_head": None,
                "matches_running_library": None, "note": "this profile predates the self-dating record"}
    rec = json.load(open(path))
    try:
        from openmm_agbnp_plugin_amd import _lib
        running = _lib.build_id()
    except Exception:  # noqa: BLE001 -- (the CPU double of the tests has no library)
        running = None
    return {"profile": os.path.relpath(d, ROOT), "library_build_id": rec.get("library_build_id"), "git_head": rec.get("git_head"),
            "taken": rec.get("taken"), "running_library_build_id": running,
            "matches_running_library": (running == rec.get("library_build_id")) if running else None}


ROW_KERNEL_NAMES = {"k_rows<0>": "k_born_rows", "k_rows<1>": "k_dborn_rows", "k_rows<2>": "k_gb_rows"}  # rocprof name -> engine name


def counter_valu_instructions(system_name, mode=None):
    """Vector wave-instructions per launch of every kernel as the SQ counters saw them (SQ_INSTS_VALU of the newest
    profiles/rNN counter summaries, taken on this workload with scripts/profile_round.sh / profile_rows.sh: the default
    configuration, the reference mode on the tile kernels, the fast mode).  None when there is none."""
    d = newest_profile_dir()
    if d is None or system_name != "1dwc":
        return None, None
    import csv
    out, used = {}, []
    files = ["fast_pmc_utilization.csv"] if mode in ("fast", "fast+single") else ["tiles_pmc_utilization.csv", "pmc_utilization.csv"]
    for name in files:  # (later files win: the default configuration's own counters over the row-form run's)
        path = os.path.join(d, name)
        if not os.path.exists(path):
            continue
        used.append(os.path.relpath(path, ROOT))
        for row in csv.DictReader(open(path)):
            try:
                out[ROW_KERNEL_NAMES.get(row["kernel"], row["kernel"])] = {
                    "valu": float(row["SQ_INSTS_VALU"]), "lds_conflict_share": float(row["lds_bank_conflict_share"]),
                    "valu_share_of_wave_cycles": float(row["valu_share_of_wave_cycles"])}
            except (KeyError, ValueError):
                continue
    return (out, " + ".join(used)) if out else (None, None)


def algorithmic_bytes(n_atoms, slots):
    """SURVEY.md s.8d:  B_alg = S*128*8 + T*4096*4 + N*104  per AGBNP1 evaluation, and its split over the
    kernels: tree sweeps 6 (cavity kernel: build W, vol-1 R, rescan R+W, vol-2 R, accumulators) + 2
    (pseudo-volume replay); one 64x64 pair-tile pass each for the 2-body
    search (cavity kernel), Born, GB and dBorn; per-atom I/O in prep/outputs."""
    nb = (n_atoms + 63) // 64
    tiles = nb * (nb + 1) // 2
    node, tile, atom = slots * 128, tiles * 4096, n_atoms * 104
    per_kernel = {
        "k_tree_cavity": 6 * node + tile,
        "k_tree_pseudo": 2 * node,
        "k_born_tiles": tile,
        "k_gb_tiles": tile,
        "k_dborn_tiles": tile,
        "k_prep": atom // 2,
        "k_outputs": atom - atom // 2,
    }
    return 8 * node + 4 * tile + atom, per_kernel


def pair_wave_steps(system, pos):
    """Wave-steps (one wave meeting 64 pairs) that the three pair kernels execute for this geometry: GB walks every
    64x64 tile of atoms; the range-limited stages walk heavy x heavy and heavy x hydrogen tiles of the pair order and
    skip tiles whose bounding boxes are more than the tables' 2 nm reach apart (same test as the kernels)."""
    n = system.n
    nb = (n + 63) // 64
    gb = (nb * (nb - 1) // 2) * 64 + nb * 32
    heavy = np.flatnonzero(system.ishydrogen == 0)
    light = np.flatnonzero(system.ishydrogen == 1)
    boxes = []
    for idx in (heavy, light):
        for b in range(0, len(idx), 64):
            p = pos[idx[b:b + 64]]
            boxes.append((p.min(axis=0), p.max(axis=0)))
    nhb = (len(heavy) + 63) // 64
    lo = np.array([b[0] for b in boxes])
    hi = np.array([b[1] for b in boxes])
    hh = hl = 0
    for i in range(nhb):
        gap = np.maximum(0.0, np.maximum(lo - hi[i], lo[i] - hi))
        near = (gap ** 2).sum(axis=1) < 4.0
        hh += 32 + 64 * int(near[i + 1:nhb].sum())  # diagonal tile: 32 wave-steps
        hl += 64 * int(near[nhb:].sum())
    return {"k_gb_tiles": {"all": gb}, "k_born_tiles": {"hh": hh, "hl": hl}, "k_dborn_tiles": {"hh": hh, "hl": hl}}


def gather_records(dist, record, device):
    """All-gather of one small python record per rank (backend nccl (= RCCL over xGMI) on GPUs, gloo in the CPU
    tests).  Returns the list of records by rank."""
    if dist is None or not dist.is_initialized():
        return [record]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, record)
    return out


def gather_throughput(dist, local_ns_day, local_ms, device):
    """All-gather of the per-replica {ns/day, ms/eval} records (2 doubles per rank).  Returns (ns_day, ms) per rank."""
    import torch
    rec = torch.tensor([local_ns_day, local_ms], dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [(float(rec[0]), float(rec[1]))]
    out = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return [(float(t[0]), float(t[1])) for t in out]


def max_over_ranks(dist, seconds, device):
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def cpu_baseline_leg(system, geometries, gpu_results, evals, version=1, **oracle_kw):
    """The ONLY place bench.py touches the oracle: the CPU restatement timed on one host core over a bounded
    sample of the same geometries.  As a by-product the sample's GPU results are compared with it."""
    from oracle import Oracle
    o = Oracle(*system.params(), version=version, **oracle_kw)
    o.execute(geometries[0])  # warm caches / page in
    t0 = time.perf_counter()
    outs = [o.execute(geometries[k]) for k in range(evals)]
    dt = time.perf_counter() - t0
    ms = 1e3 * dt / evals
    de = max(abs(outs[k][0] - gpu_results[k][0]) for k in range(min(evals, len(gpu_results))))
    df = max(float(np.abs(outs[k][1] - gpu_results[k][1]).max()) for k in range(min(evals, len(gpu_results))))
    return ms, de, df


def load_workload(name):
    if name.endswith("_x4"):  # HIV-RT stand-in (BASELINE.json config 4): 2x2x1 lattice of copies, 7 nm pitch
        return P.lattice(P.load_system(name[:-3]), 2, 2, 1, 7.0)
    return P.load_system(name)


class NotSettled(SystemExit):
    """A pass whose evaluations kept being withheld.  Carries what the engine said about it (Replica.report)."""

    def __init__(self, what, report=None):
        self.report = report or {}
        h = self.report.get("history") or [{}]
        last = h[-1]
        detail = (f"{self.report.get('phase', '?')} try {self.report.get('tries', '?')}; last withheld {last.get('withheld', '?')} "
                  f"kinds {last.get('overflow_kinds', '?')} level {last.get('pack_level', '?')} forests {last.get('forests', '?')} "
                  f"variant {last.get('variant', '?')} indices {last.get('withheld_indices', '?')}")
        super().__init__(f"bench: tree capacity did not settle ({what}): {detail}")


class HipBackend:
    """What a worker asks of torch.cuda and of the engine, in one place: one GPU of the node (LOCAL_RANK), device buffers,
    the stream, the engine's kernel object.  The CPU tests of the multi-process path swap it for a double of the same shape
    (AGBNP_BENCH_BACKEND_MODULE=<module with a Backend class>): everything else of the worker -- the collectives, the
    retry / abort decisions, the JSON line -- runs as it does on GPUs."""

    name = "hip"

    def __init__(self, torch, local_rank, collective_backend):
        self.torch = torch
        # one process per GPU.  AGBNP_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks then
        # share devices (local rank modulo the device count) and the collectives run on CPU tensors.
        self.index = local_rank % max(torch.cuda.device_count(), 1) if collective_backend != "nccl" else local_rank
        torch.cuda.set_device(self.index)
        self.device = torch.device("cuda", self.index)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def wait_idle(self):
        """synchronize(), reached through a spin on the stream's status: a blocking wait sleeps in the driver and comes back
        tens of microseconds after the last kernel has ended -- at the driver's 20 steps that is a microsecond per step of
        host latency, not of evaluation.  (The synchronize behind the spin returns at once and keeps the contract literal.)"""
        st = self.torch.cuda.current_stream()
        while not st.query():
            pass
        self.torch.cuda.synchronize()

    def current_stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def new_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def stream_scope(self, stream):
        return self.torch.cuda.stream(stream)

    def tensor(self, array, dtype):
        return self.torch.tensor(array, dtype=dtype, device=self.device).contiguous()

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype, device=self.device)

    def random_walk(self, start, steps, sigma, seed):
        """[steps, n, 3] positions of a cumulative random walk from `start` (independent N(0, sigma) steps per coordinate),
        made on the device."""
        torch = self.torch
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        inc = torch.randn((steps,) + tuple(start.shape), dtype=torch.float64, device=self.device, generator=g) * sigma
        return (torch.cumsum(inc, dim=0) + self.tensor(start, torch.float64)).contiguous()

    def identity(self):
        props = self.torch.cuda.get_device_properties(self.index)
        return {"device_index": self.index, "device_name": props.name, "device_uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": int(getattr(props, "pci_bus_id", -1))}

    def kernel(self, mode=None):
        return P.HipCalcAGBNPForceKernel(device=self.index, mode=mode) if mode else P.HipCalcAGBNPForceKernel(device=self.index)


class Replica:
    """One context + its device-resident geometries, forces and energy."""

    def __init__(self, dev, system, version, steps, seed0, method=None, cutoff=1.0, mode=None, stream=None, geometries=None):
        torch = dev.torch
        self.dev, self.system, self.n = dev, system, system.n
        force = P.AGBNPForce.from_arrays(*system.params(), version=version)
        force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic if method is None else method)  # example/1dwc_benchmark.py:10
        force.setCutoffDistance(cutoff)
        self.kernel = dev.kernel(mode)
        self.kernel.initialize(force)
        if geometries is None:
            self.geoms = np.stack([system.jittered(seed0 + s) for s in range(steps)])
            self.d_pos = dev.tensor(self.geoms, torch.float64)
        else:  # (device-resident already: the random walk of the drift record)
            self.geoms, self.d_pos = None, geometries
        self.d_force = dev.zeros((self.n, 3), torch.float64)
        self.d_energy = dev.zeros((1,), torch.float64)
        self.stream = dev.current_stream() if stream is None else stream.cuda_stream
        self.step_bytes = self.n * 3 * 8
        self.n_geoms = int(self.d_pos.shape[0])
        # AGBNP_BENCH_FAIL_AT=<rank>:<k>: the k-th evaluation of that rank's TIMED pass raises (test hook, see Replica.run)
        want = os.environ.get("AGBNP_BENCH_FAIL_AT", "").split(":")
        self.fail_at = int(want[1]) if len(want) == 2 and int(want[0]) == int(os.environ.get("RANK", "0")) else -1
        self.in_timed_pass, self.timed_calls = False, 0

    def run(self, first, count):
        base = self.d_pos.data_ptr()
        for s in range(first, first + count):
            if self.in_timed_pass:  # (tests of the N > 1 path on real GPUs: a rank that dies between the barriers)
                self.timed_calls += 1
                if self.timed_calls == self.fail_at:
                    raise RuntimeError(f"injected failure at evaluation {self.timed_calls} of this rank's timed pass (AGBNP_BENCH_FAIL_AT)")
            self.kernel.execute_device(base + s * self.step_bytes, self.d_force.data_ptr(), self.d_energy.data_ptr(), self.stream)

    def state(self):
        """What the engine says about its capacity negotiation right now (for the message of a pass that would not settle):
        why the last finish() withheld evaluations (scalar 15: 1 nodes, 2 local atoms, 4 packing, 8 rows, 16 order / masks,
        32 / 64 a forest's nodes / atoms, 256.. the part count of a lone item), the packing level, forests, store variant and
        the first withheld indices."""
        out = {}
        for name in ("overflow_kinds", "pack_level", "forests", "variant", "pack_plans", "max_subtree_nodes"):
            try:
                out[name] = int(self.kernel.scalar(name))
            except Exception:  # noqa: BLE001 -- (the CPU doubles of the tests answer fewer names)
                pass
        try:
            out["withheld_indices"] = [int(i) for i in list(self.kernel.withheld())[:8]]
        except Exception:  # noqa: BLE001
            pass
        return out

    def settle(self, count, agree=None):
        """Warm-up: also settles the tree-capacity variant (repeat until no evaluation of the batch was withheld -- on ANY
        rank: `agree` turns the local count into the job's, so every rank repeats or none does).  The batch of a repeat starts
        one geometry later than the one before (the window slides): a repeat is not a replay of the same sequence from the
        same state."""
        agree = agree or (lambda withheld: withheld)
        self.report = {"phase": "settle", "tries": 0, "history": []}
        span = max(1, getattr(self, "n_geoms", max(count, 1)) - max(count, 1) + 1)
        for k in range(8):
            self.report["tries"] = k + 1
            self.run(k % span, max(count, 1))
            local = self.kernel.finish(self.stream)
            if local:
                self.report["history"].append(dict(self.state(), withheld=int(local), try_number=k + 1))
            if not agree(local):
                return True
        return False

    def preheat(self, seconds, batch, evals=0):
        """Evaluations until the device has been busy for `seconds` (the headline: the first milliseconds after an idle period
        run at a lower clock -- the driver's 20-step / 5-warm-up invocation measured 0.1216 ms where 200 steps measured 0.1147)
        or, with `evals`, exactly that many (the secondary records: the phase of the packing's plans in which their timed pass
        starts must not depend on the box's speed).  Untimed; returns the number of evaluations."""
        done = 0
        t0 = time.perf_counter()
        while (done < evals) if evals else (time.perf_counter() - t0 < seconds):
            self.run(0, max(batch, 1))
            self.dev.synchronize()
            done += max(batch, 1)
        self.kernel.finish(self.stream)  # (a withheld one here is caught by the timed pass's own check)
        return done

    RECOVER_EVALS = 80  # (more than the 64 evaluations after which the packing gives a tightened level back)

    def recover(self, first, count):
        """Between two tries of a timed pass: untimed evaluations over the pass's own geometries until the engine has had its
        relax period, so that the next try does not start from the state the failed one left (identity packing, a raised
        level) and walk the same plan sequence over the same geometries again."""
        done = 0
        while done < self.RECOVER_EVALS:
            step = min(max(count, 1), self.RECOVER_EVALS - done)
            self.run(first, step)
            done += step
        self.kernel.finish(self.stream)

    def timed(self, first, count, barrier=lambda: None, agree=None, tries=4):
        """Seconds for `count` evaluations, or None if they would not settle.  `agree` (multi-rank runs): the MAX over the
        ranks of the withheld count, a collective that EVERY rank enters after every try -- a rank whose own evaluations were
        all complete repeats with the others instead of walking on to the next collective alone."""
        agree = agree or (lambda withheld: withheld)
        self.tries = 0
        self.report = {"phase": "timed", "tries": 0, "history": []}
        for k in range(tries):
            self.tries += 1
            self.report["tries"] = self.tries
            self.d_force.zero_()
            self.d_energy.zero_()
            self.dev.synchronize()
            barrier()
            self.dev.synchronize()
            t0 = time.perf_counter()
            self.in_timed_pass = True
            try:
                self.run(first, count)
            finally:
                self.in_timed_pass = False
            self.dev.wait_idle()
            t1 = time.perf_counter()  # (this rank's time; the job's is the MAX over the ranks, taken by the caller)
            barrier()
            # finish() reads the device's sticky overflow log: EVERY one of the timed evaluations is accounted for, not
            # just the last.  Non-zero = some were withheld (capacity negotiation still under way): time the run again.
            local = self.kernel.finish(self.stream)
            if local:
                self.report["history"].append(dict(self.state(), withheld=int(local), try_number=self.tries))
            if not agree(local):
                return t1 - t0
            if k + 1 < tries:
                self.recover(first, count)  # (every rank: the verdict was the job's)
        return None

    def host_results(self, first, count):
        out = []
        for k in range(count):
            f = np.zeros((self.n, 3))
            e = self.kernel.execute(self.geoms[first + k], f)
            out.append((e, f))
        return out


def secondary_entry(dev, name, version, steps, warmup, cpu_evals, method=None, cutoff=1.0, mode=None, preheat_evals=320, **oracle_kw):
    """ms/eval, ns/day and parity-on-sample of another configuration (rank 0, one GPU); the headline's protocol: settle,
    a short pre-heat (a new context starts on cold caches and an unplanned forest packing; by COUNT, so that the plan phase
    the timed pass starts in is the same on every box), then the timed steps."""
    system = load_workload(name)
    r = Replica(dev, system, version, steps + warmup, 7000, method=method, cutoff=cutoff, mode=mode)
    if not r.settle(warmup):
        raise NotSettled(name, r.report)
    r.preheat(0.0, max(warmup, 5), evals=preheat_evals)
    seconds = r.timed(warmup, steps)
    if seconds is None:
        raise NotSettled(name, r.report)
    ms = 1e3 * seconds / steps
    entry = {"workload": name, "atoms": system.n, "version": version, "ms_per_eval": ms, "ns_day": 86.4 / ms, "timed_tries": r.tries,
             "forests": int(r.kernel.scalar("forests")), "pack_level": int(r.kernel.scalar("pack_level")), "kernel_variant": int(r.kernel.scalar("variant"))}
    if r.report.get("history"):
        entry["withheld_tries"] = r.report["history"]
    if cpu_evals > 0:
        cpu_ms, de, df = cpu_baseline_leg(system, r.geoms[warmup:], r.host_results(warmup, cpu_evals), cpu_evals, version=version, **oracle_kw)
        entry.update({"cpu_ms_per_eval": cpu_ms, "parity_on_sample": {"evals": cpu_evals, "max_abs_dE_kJmol": de, "max_abs_dF_kJmolnm": df}})
    return entry


def _with_env(overrides, fn):
    """fn() under environment overrides (the engine reads its knobs when a context is created)."""
    saved = {k: os.environ.get(k) for k in overrides}
    os.environ.update(overrides)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def rebuild_entry(dev, name, steps, warmup):
    """What the headline's protocol never contains: an evaluation that REBUILDS the neighbour rows of the row-form pair
    stages.  The rows carry a 0.1 nm skin and are rebuilt (on the device, inside the evaluation) when an atom has moved more
    than half of it; the headline's jittered geometries never do.  Here the same workload runs in a context whose move
    threshold is zero (AGBNP_HIP_ROW_MOVE=0: same skin, same list lengths, a rebuild at every new geometry) next to a plain
    context on the same box; the difference is the price of one rebuild."""
    system = load_workload(name)

    def timed(env):
        def go():
            r = Replica(dev, system, 1, steps + warmup, 8000)
            if not r.settle(warmup):
                raise NotSettled("rebuild record", r.report)
            b0 = int(r.kernel.scalar("row_builds"))
            seconds = r.timed(warmup, steps)
            if seconds is None:
                raise NotSettled("rebuild record", r.report)
            return 1e3 * seconds / steps, int(r.kernel.scalar("row_builds")) - b0, int(r.kernel.scalar("rows_on"))
        return _with_env(env, go)

    plain_ms, plain_builds, rows_on = timed({})
    if not rows_on:
        return {"workload": name, "rows_on": 0}
    rebuild_ms, builds, _ = timed({"AGBNP_HIP_ROW_MOVE": "0"})
    return {"workload": name, "steps": steps, "plain_eval_ms": plain_ms, "builds_in_plain_timed_region": plain_builds,
            "rebuild_eval_ms": rebuild_ms, "builds_in_rebuild_timed_region": builds, "rebuild_cost_ms": rebuild_ms - plain_ms,
            "note": "AGBNP_HIP_ROW_MOVE=0 (rows rebuilt at every new geometry, default skin) against the default context, same box, same geometries"}


def drift_entry(dev, name, steps, sigma, warmup=20):
    """ms per evaluation along a CUMULATIVE random walk (every coordinate of every atom moves by N(0, sigma) per step, no
    tethers, no force field): atoms drift out of the neighbour rows' skin, so rebuild evaluations -- and the extra
    slice-tuning rebuild, and the forest re-plans -- fall INSIDE the timed region and are counted.  A random walk drifts
    faster than thermal motion (which oscillates): an upper bound on the rebuild rate of real MD at that step length."""
    system = load_workload(name)
    torch = dev.torch
    walk = dev.random_walk(system.pos, steps + warmup, sigma, 20261004)
    r = Replica(dev, system, 1, steps + warmup, 0, geometries=walk)
    if not r.settle(warmup):
        raise NotSettled("drift record", r.report)
    k = r.kernel
    b0, p0 = int(k.scalar("row_builds")), int(k.scalar("pack_plans"))
    # ONE pass.  The overflow log is read every 500 evaluations (it names 2048); evaluations it names as withheld -- a drifting
    # geometry outgrows the forest packing planned a few steps earlier now and then -- are REPEATED at once, inside the timed
    # region, as an MD driver must (the walk itself does not depend on the forces, so the repeat can run out of order)
    r.d_force.zero_()
    r.d_energy.zero_()
    dev.synchronize()
    t0 = time.perf_counter()
    withheld, repeats, chunk = 0, 0, 500
    for first in range(warmup, warmup + steps, chunk):
        todo = list(range(first, min(first + chunk, warmup + steps)))  # the steps still owed
        for attempt in range(8):
            for step in todo:
                r.run(step, 1)
            bad = r.kernel.finish(r.stream)
            if not bad:
                break
            named = r.kernel.withheld()  # (numbered from the first evaluation enqueued since the finish before)
            if len(named) != bad:
                raise SystemExit("bench: the overflow log does not name every withheld evaluation of the drift record")
            todo = [todo[i] for i in named]
            withheld += bad
            repeats += len(todo)
        else:
            raise SystemExit("bench: the repeats of the drift record did not converge")
    dev.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    disp = (walk[-1] - walk[warmup]).norm(dim=1)
    return {"workload": name, "steps": steps, "sigma_step_nm": sigma, "ms_per_eval": ms, "ns_day": 86.4 / ms,
            "builds_in_timed_region": int(k.scalar("row_builds")) - b0, "forest_plans_in_timed_region": int(k.scalar("pack_plans")) - p0,
            "withheld_evaluations": int(withheld), "repeated_evaluations_in_timed_region": int(repeats),
            "rms_displacement_nm": float(torch.sqrt((disp ** 2).mean())),
            "max_displacement_nm": float(disp.max()), "kernel_variant_at_end": int(k.scalar("variant")),
            "total_nodes_at_end": int(k.scalar("total_nodes")), "max_subtree_nodes_at_end": int(k.scalar("max_subtree_nodes")),
            "note": "cumulative random walk from the file coordinates, no tethers; rows rebuilt on the device when an atom has moved "
                    "more than half the 0.1 nm skin; finish() every 500 evaluations and the repeats of withheld evaluations inside the timed region"}


def openmm_entry(dev, name, steps, warmup, shuffled=True):
    """ms per evaluation through agbnp_hip_execute_openmm (an OpenMM GPU context's conventions: posq in the context's atom
    order + atomIndex, fixed-point force planes, energy buffer) under the host protocols of the glue: a blocking finish()
    after every evaluation (the reference's own protocol: the stream is drained), wait_verdict() after every evaluation
    (as strict -- the host learns whether THIS evaluation was withheld before it goes on -- but it waits for a pinned word
    the device writes when the tree stage has ended, not for the stream), a non-blocking poll() after every evaluation
    (finish only on demand; a withheld evaluation is found late), and a finish every 64 evaluations."""
    torch = dev.torch
    system = load_workload(name)
    n = system.n
    padded = (n + 31) // 32 * 32
    force = P.AGBNPForce.from_arrays(*system.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    kernel = dev.kernel()
    kernel.initialize(force)
    rng = np.random.default_rng(3)
    # context slot -> particle.  OpenMM keeps its atoms in a spatially local order (reorderAtoms); a random permutation is the worst
    # case for the fixed-point force planes (every add of a forest's atoms lands on a cache line of its own), particle order the
    # friendly one (neighbours in the file are neighbours in space): both are reported
    perm = rng.permutation(n) if shuffled else np.arange(n)
    geoms = [system.jittered(5000 + k) for k in range(steps + warmup)]
    posq = np.zeros((len(geoms), padded, 4))
    for k, g in enumerate(geoms):
        posq[k, :n, :3] = g[perm]
    d_posq = dev.tensor(posq, torch.float64)
    d_index = dev.tensor(np.concatenate([perm, np.arange(n, padded)]), torch.int32)
    d_force = dev.zeros((3 * padded,), torch.int64)
    d_energy = dev.zeros((64,), torch.float64)
    stream = dev.current_stream()
    step_bytes = padded * 4 * 8

    def run(k):
        kernel.execute_openmm(d_posq.data_ptr() + k * step_bytes, True, 0, d_index.data_ptr(), padded, d_force.data_ptr(), d_energy.data_ptr(), True, 0, stream)

    for k in range(warmup):
        run(k)
    kernel.finish(stream)
    out = {"workload": name, "entry_point": "agbnp_hip_execute_openmm (double precision context, " + ("shuffled atom order)" if shuffled else "particle order)")}
    for label, after in (("finish_every_evaluation", lambda k: kernel.finish(stream)),
                         ("wait_verdict_every_evaluation", lambda k: kernel.wait_verdict()[1] and kernel.finish(stream)),
                         ("poll_every_evaluation", lambda k: kernel.poll()[1] and kernel.finish(stream)),
                         ("finish_every_64", lambda k: (k % 64 == 63) and kernel.finish(stream))):
        dev.synchronize()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            run(k)
            after(k)
        dev.synchronize()
        out["ms_per_eval_" + label] = 1e3 * (time.perf_counter() - t0) / steps
        kernel.finish(stream)
    out["launches_per_evaluation"] = int(kernel.scalar("launches"))
    # where the entry point's time goes: raw hipEvent intervals per kernel (each holds one event pair's overhead, the same on both
    # sides) through this entry point and, same context, same geometries in particle order, through agbnp_hip_execute_device
    d_pos = dev.tensor(np.stack(geoms), torch.float64)
    d_f64 = dev.zeros((n, 3), torch.float64)
    d_e64 = dev.zeros((1,), torch.float64)

    def raw_kernel_us(run_one):
        for k in range(warmup, warmup + 8):  # (a switch of the entry point rewrites the roots' position words once)
            run_one(k)
        kernel.finish(stream)
        kernel.set_profiling(True)
        for k in range(warmup, warmup + steps):
            run_one(k)
        kernel.finish(stream)
        times = kernel.kernel_times()
        kernel.set_profiling(False)
        return {name: round(1e3 * v[0] / max(v[1], 1), 2) for name, v in times.items() if v[1] > 0}

    out["kernel_event_us"] = raw_kernel_us(run)
    out["kernel_event_us_execute_device_same_context"] = raw_kernel_us(
        lambda k: kernel.execute_device(d_pos.data_ptr() + k * n * 24, d_f64.data_ptr(), d_e64.data_ptr(), stream))
    return out


def md_loop_entry(dev, name, steps):
    """ms per MD step of the loop behind examples/1dwc_benchmark.py (the py3 counterpart of the script the reference's
    ns/day comes from, example/1dwc_benchmark.py:20,29-33): Langevin 300 K, 1 / ps, 1 fs, AGBNP1 + tethers, one step = two
    integrator launches (csrc/md_kernels.hip) around one evaluation, captured once as a HIP graph and replayed; the host
    reads the overflow log every 1000 steps.  Example support, NOT a full force field (no bonded or nonbonded terms): the
    figure is comparable to nothing the reference produces; the headline `value` stays the force-limited figure."""
    from openmm_agbnp_plugin_amd.md import DeviceMD
    system = load_workload(name)
    force = P.AGBNPForce.from_arrays(*system.params(), version=1)
    force.setNonbondedMethod(P.AGBNPForce.CutoffNonPeriodic)
    kernel = dev.kernel()
    kernel.initialize(force)
    md = DeviceMD(system, kernel, k_tether=1.0e5, dt=0.001, temperature=300.0, friction=1.0, device=f"cuda:{dev.index}")
    md.settle()
    md.forces()
    kernel.finish()
    md.run(20, "langevin", check_every=20)
    dev.synchronize()
    t0 = time.perf_counter()
    missed = md.run(steps, "langevin", check_every=1000)
    dev.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {"workload": name, "script": "examples/1dwc_benchmark.py", "steps": steps, "ms_per_step": ms, "ns_day": 86.4 / ms,
            "steps_without_agbnp_term": int(missed)}


def concurrent_replicas_entry(dev, name, replicas, steps, warmup):
    """Aggregate throughput of several INDEPENDENT replicas sharing one GPU, each context on a stream of its own (multiple
    walkers / replica exchange on one device).  The evaluation is bound by dependent latency, not by throughput, so
    kernels of different replicas overlap; this is NOT the headline metric (one replica per GPU), it shows the headroom."""
    system = load_workload(name)
    streams = [dev.new_stream() for _ in range(replicas)]
    reps = []
    for r, st in enumerate(streams):
        with dev.stream_scope(st):
            rep = Replica(dev, system, 1, steps + warmup, 9000 + 977 * r, stream=st)
            if not rep.settle(warmup):
                raise NotSettled("concurrent replicas", rep.report)
            reps.append(rep)
    dev.synchronize()
    for attempt in range(3):
        t0 = time.perf_counter()
        for s in range(warmup, warmup + steps):  # round-robin enqueue: one evaluation of every replica per turn
            for rep in reps:
                rep.run(s, 1)
        dev.synchronize()
        t1 = time.perf_counter()
        if not any([rep.kernel.finish(rep.stream) for rep in reps]):  # (a list: every replica's log is read and reset)
            break
    else:
        raise SystemExit("bench: the concurrent replicas' third attempt still held withheld evaluations")
    ms = 1e3 * (t1 - t0) / steps  # per round of `replicas` evaluations
    return {"workload": name, "replicas_on_one_gpu": replicas, "ms_per_round": ms, "ms_per_eval_aggregate": ms / replicas,
            "aggregate_ns_day": replicas * 86.4 / ms}


class Line:
    """The ONE JSON line of the contract.  Once the headline exists the line is ALWAYS printed, exactly once: `emit` is called
    from a `finally`, the optional records add to it through `put` / `append` / `update` (under a lock), a record that fails is
    an {"error": ...} entry of its own (`optional`), and a watchdog prints the line with what it has -- and ends the process with
    exit code 0 -- when the optional records overrun their time budget (a hang in one of them must not cost the line either)."""

    def __init__(self, result):
        import threading
        self.result, self.lock, self.printed, self.timer = result, threading.Lock(), False, None

    def put(self, key, value):
        with self.lock:
            self.result[key] = value

    def append(self, key, value):
        with self.lock:
            self.result[key].append(value)

    def update(self, key, values):
        with self.lock:
            self.result[key].update(values)

    def optional(self, name, fn, *a, **kw):
        try:
            if os.environ.get("AGBNP_BENCH_FAIL_RECORD") == name:  # (test hook: this record fails, whatever it would have measured)
                raise SystemExit(f"injected failure of the record '{name}' (AGBNP_BENCH_FAIL_RECORD)")
            return fn(*a, **kw)
        except KeyboardInterrupt:
            raise
        except BaseException as exc:  # noqa: BLE001 -- SystemExit included: an optional record never ends the job
            entry = {"error": f"{type(exc).__name__}: {exc}"}
            if getattr(exc, "report", None):
                entry["report"] = exc.report
            print(f"bench: record '{name}' failed and is reported inside the line: {entry['error']}", file=sys.stderr, flush=True)
            return entry

    def emit(self, note=None):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if note:
                self.result["optional_records_aborted"] = note
            if self.timer is not None:
                self.timer.cancel()
            print(json.dumps(self.result), flush=True)

    def watchdog(self, seconds):
        import threading

        def fire():
            print(f"bench: the optional records overran their budget of {seconds:.0f} s: the line is printed with what it has", file=sys.stderr, flush=True)
            self.emit(note=f"watchdog after {seconds:.0f} s")
            sys.stderr.flush()
            os._exit(0)

        if seconds > 0:
            self.timer = threading.Timer(seconds, fire)
            self.timer.daemon = True
            self.timer.start()


class JobAborted(SystemExit):
    """Raised by Job.abort: the job has ended on every rank (never caught by Job.run)."""


class Job:
    """Multi-rank discipline.  The ranks run different geometries, so anything can go wrong on one rank alone: an evaluation
    withheld, a capacity that does not settle, a HIP error.  EVERY collective of the job is therefore the SAME operation --
    `sync`: the MAX-all-reduce of the three doubles {failed, withheld, seconds} -- entered by every rank at the same points
    of the program (the barriers around the timed region, the verdict after every try, the end of every phase), and every
    one of them carries the failure flag.  A rank that fails anywhere (its phase runs under `run`, which catches the
    exception) raises the flag in the NEXT collective it enters, which is the next collective every other rank enters, of
    whatever purpose; every rank that sees the flag leaves at once through `abort` without entering another one.  So the
    k-th collective of one rank always meets the k-th of every other, and nobody waits in a collective that the others never
    enter.  (Round 3 had a barrier, a withheld vote and a failure vote as three different operations: a failure vote could
    be consumed as a withheld vote.)  One rank: the same calls, no communication."""

    def __init__(self, dist, device, rank):
        self.dist, self.device, self.rank, self.error = dist, device, rank, None
        self.multi = dist is not None and dist.is_initialized()  # (a one-rank group too: the rehearsal of the backend)

    def sync(self, withheld=0, seconds=0.0):
        """-> (MAX withheld, MAX seconds) over the ranks; leaves through abort() if any rank has failed."""
        vec = [1.0 if self.error is not None else 0.0, float(withheld), float(seconds)]
        if self.multi:
            import torch
            t = torch.tensor(vec, dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            vec = [float(v) for v in t.tolist()]
        if vec[0] > 0.0:
            self.abort()
        return int(vec[1]), vec[2]

    def barrier(self):
        self.sync()

    def agree(self, withheld):
        """The job's verdict on a try: how many evaluations the worst rank had withheld."""
        return self.sync(withheld=1 if withheld else 0)[0]

    def run(self, what, fn, *a, **kw):
        """fn() of one phase; a failure is kept for the next collective (which ends the job on every rank)."""
        if self.error is not None:
            return None
        try:
            return fn(*a, **kw)
        except JobAborted:  # (a collective inside the phase saw another rank's failure: the job is over, no more collectives)
            raise
        except (SystemExit, Exception) as exc:  # noqa: BLE001 -- the failure is reported, then every rank leaves together
            self.error = f"{what}: {type(exc).__name__}: {exc}"
            return None

    def abort(self):
        if self.rank == 0 or self.error is not None:
            print(f"bench: rank {self.rank}: " + (f"failed: {self.error}" if self.error else "another rank failed") + "; the job ends on every rank",
                  file=sys.stderr, flush=True)
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
        raise JobAborted(1)


def headline_pass(rep, job, W, K, preheat_seconds):
    """Settle, pre-heat and time the K steps of this rank's replica; every repeat decision is the job's (Job.agree), so all
    ranks make the same number of tries.  Returns (seconds of this rank, evaluations spent on the pre-heat)."""

    def settle():
        if not rep.settle(W, job.agree):
            raise NotSettled("the headline's warm-up", rep.report)
        return rep.preheat(preheat_seconds, max(W, 5))

    warm = job.run("warm-up", settle)
    job.sync()

    def timed():
        seconds = rep.timed(W, K, job.barrier, job.agree)
        if seconds is None:
            raise NotSettled("the headline's timed pass", rep.report)
        return seconds

    seconds = job.run("timed pass", timed)
    job.sync()
    return seconds, warm


def make_backend(torch, local_rank, collective_backend):
    module = os.environ.get("AGBNP_BENCH_BACKEND_MODULE")  # (CPU tests of the multi-process path: a double of HipBackend)
    if module:
        return importlib.import_module(module).Backend(torch, local_rank, collective_backend)
    return HipBackend(torch, local_rank, collective_backend)


def worker(args):
    """One rank = one replica on one GPU (the whole job when --gpus 1)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("AGBNP_BENCH_BACKEND", "nccl")
    dev = make_backend(torch, local_rank, backend)
    sys.stderr.write(f"bench: rank {rank} of {env_world}: pid {os.getpid()}, device {getattr(dev, 'index', '?')}\n")  # (one write: the ranks share the pipe)
    sys.stderr.flush()
    coll_device = dev.device if backend == "nccl" else torch.device("cpu")
    # AGBNP_BENCH_FORCE_DIST=1: a ONE-rank job still initialises the process group and sends every collective through the
    # backend (a rehearsal of the RCCL plumbing on a box with one GPU)
    force_dist = os.environ.get("AGBNP_BENCH_FORCE_DIST", "0") not in ("", "0")
    if env_world > 1 or force_dist:
        if force_dist and env_world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev.device)
        else:
            dist.init_process_group(backend=backend)
    world = dist.get_world_size() if dist.is_initialized() else 1  # what took part, not what the environment promised
    mdist = dist if (world > 1 or force_dist) else None
    job = Job(mdist, coll_device, rank)

    K, W = args.steps, args.warmup
    mode = None if args.mode == "reference" else args.mode
    system = job.run("workload", load_workload, args.system)
    rep = job.run("context creation", lambda: Replica(dev, system, 1, K + W, 1000 * rank, mode=mode))
    job.sync()
    n = system.n
    kernel = rep.kernel
    elapsed_local, warm_evals = headline_pass(rep, job, W, K, args.preheat_ms * 1e-3)
    _, elapsed = job.sync(seconds=elapsed_local)  # the job's time is its slowest rank's
    ms_per_step = 1e3 * elapsed / K
    record = {"rank": rank, "local_rank": local_rank, **dev.identity(),
              "ms_per_eval": 1e3 * elapsed_local / K, "ns_day": 86.4 / (1e3 * elapsed_local / K), "pid": os.getpid(),
              "timed_tries": rep.tries, "clock_warm_evals": warm_evals}
    if mdist is not None and args.cpu_evals > 0:
        # N > 1: EVERY rank checks the first of its own timed geometries against the CPU oracle (the ranks run different
        # geometries on different devices; rank 0's sample says nothing about rank 5's GPU).  All ranks at once, one host
        # core each, outside the timed region; a rank that disagrees ends the job.
        def own_sample():
            oracle_kw = {"cutoff": 1.0} if mode in ("fast", "fast+single") else {}
            cpu_ms, de, df = cpu_baseline_leg(system, rep.geoms[W:], rep.host_results(W, 1), 1, **oracle_kw)
            if mode != "fast+single" and not (de < 1e-4 and df < 1e-4):
                raise SystemExit(f"rank {rank}: the HIP path disagrees with the CPU oracle on this rank's sample (dE {de:.3e}, dF {df:.3e})")
            return {"evals": 1, "max_abs_dE_kJmol": de, "max_abs_dF_kJmolnm": df, "cpu_ms_per_eval": cpu_ms}
        record["parity_on_sample"] = job.run("parity sample of this rank", own_sample)
        job.sync()
    per_rank = gather_records(mdist, record, coll_device)
    value = world * 86.4 / ms_per_step  # whole job: all replicas' steps / max-over-ranks time

    # ---- secondary aggregate: R concurrent replicas per GPU (every rank, same R; the job's time is the slowest rank's)
    per_gpu = None
    if args.replicas_per_gpu > 1:
        local = job.run("concurrent replicas per GPU", concurrent_replicas_entry, dev, args.system, args.replicas_per_gpu, K, W)
        _, ms_round = job.sync(seconds=local["ms_per_round"] if local else 0.0)
        per_gpu = {"replicas_per_gpu": args.replicas_per_gpu, "ms_per_round_max_over_ranks": ms_round,
                   "aggregate_ns_day": world * args.replicas_per_gpu * 86.4 / ms_round,
                   "note": "R independent contexts per GPU on R streams of one process; NOT the headline (one replica per GPU)"}

    # ---- the closing collectives: from here on rank 0 works alone (per-kernel pass, CPU baseline, secondary records) and
    #      the other ranks are done -- nobody sits in a collective while rank 0 runs twenty seconds of CPU oracle
    if mdist is not None:
        job.sync()
        dist.destroy_process_group()
    if rank != 0:
        return 0

    slots = int(kernel.scalar("total_nodes")) + (n - system.nheavy) + 1  # + hydrogen slots + root, as the reference counts
    b_eval, b_kernel = algorithmic_bytes(n, slots)
    semantics = {None: "Reference semantics: all pairs",
                 "fast": "FAST mode: OpenCL-platform semantics, every pair stage truncated at the cutoff",
                 "fast+single": "FAST mode: OpenCL-platform semantics, every pair stage truncated at the cutoff; GB pair terms in packed FP32",
                 "deterministic": "Reference semantics: all pairs; DETERMINISTIC mode: quantized order-dependent sums"}[mode]
    result = {
        "metric": "AGBNP1 force-eval-limited ns/day (1 fs step), thrombin 1dwc, independent replicas",
        "value": value, "unit": "ns/day", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "force_eval_ms": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64 (GB pair terms f32)" if mode == "fast+single" else "f64", "data": "synthetic",
        "config": {"workload": f"{args.system} ({'thrombin, ' if args.system == '1dwc' else ''}{n} atoms, {system.nheavy} heavy) AGBNP1 version=1, "
                               f"CutoffNonPeriodic 1.0 nm ({semantics}), one jittered geometry per step "
                               "(sigma 0.002 nm), positions/forces/energy resident in HBM",
                   "replicas": world, "tree_slots": slots, "kernel_variant": int(kernel.scalar("variant")), "mode": args.mode,
                   "pair_stage_form": "rows" if int(kernel.scalar("rows_on")) else "tiles",
                   "forests": int(kernel.scalar("forests")), "pack_level": int(kernel.scalar("pack_level"))},
        "clock_warm_evals": warm_evals,
        "per_replica_ns_day": [round(r["ns_day"], 4) for r in per_rank],
        "ranks": per_rank,
        "distinct_devices": len({(r["device_uuid"], r["pci_bus_id"], r["device_index"]) for r in per_rank}),
        "launcher": os.environ.get("AGBNP_BENCH_LAUNCHER", "torch.distributed.run" if env_world > 1 else "none"),
        "collectives": (backend if mdist is not None else "none (one rank)"),
        "algorithmic_bytes_per_eval": b_eval,
        "eval_hbm_fraction": (b_eval / (ms_per_step * 1e-3)) / (HBM_PEAK_GBS * 1e9),
    }
    if per_gpu is not None:
        result["replicas_per_gpu"] = per_gpu
    if int(kernel.scalar("rows_on")):
        result["neighbour_rows"] = {"builds_so_far": int(kernel.scalar("row_builds")),
                                    "entries_per_slice": int(kernel.scalar("row_slice")),
                                    "note": "rows built with a skin and rebuilt on the device when an atom has moved more than half of it; "
                                            "the jittered geometries of the headline protocol stay within it (no rebuild inside its timed "
                                            "region): rebuild_eval_ms and the drift record price what it leaves out"}

    def roofline_section():
        # ---- per-kernel durations: same K steps again with a hipEvent in front of every kernel (separate pass so
        #      that the events do not sit inside the timed region above)
        kernel.set_profiling(True)
        rep.run(W, K)
        if kernel.finish(rep.stream):
            print("bench: an evaluation of the profiling pass overflowed after the timed pass had settled", file=sys.stderr)
        times = kernel.kernel_times()
        kernel.set_profiling(False)
        raw_us = {k: 1e3 * v[0] / max(v[1], 1) for k, v in times.items() if v[1] > 0}
        # An interval between two event records holds one kernel plus the cost of the event pair (~2.5 us here).
        # Without the events the launches run back to back (rocprofv3 trace: < 0.1 us between kernels), so the timed
        # step is the sum of the kernel durations: the same per-interval overhead is taken off every kernel such that
        # the sum closes on the measured step time.  (The raw figures are kept in kernel_event_us; the independent check of
        # the split is the rocprofv3 summary of the same command under profiles/.)
        event_overhead_us = max(0.0, (sum(raw_us.values()) - 1e3 * ms_per_step) / max(len(raw_us), 1))
        avg_us = {k: max(v - event_overhead_us, 0.0) for k, v in raw_us.items()}
        dominant = max(avg_us, key=avg_us.get)
        traffic, eval_traffic = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_pmc.json")
        if os.path.exists(tfile):
            try:
                rec = json.load(open(tfile))
                if rec.get("system") == args.system:
                    eval_traffic = rec.get("all_kernels_bytes_per_eval")
                    if rec.get("kernel") == dominant:
                        traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        achieved = b_kernel.get(dominant, 0) / (avg_us[dominant] * 1e-6) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "algorithmic_bytes_per_launch": b_kernel.get(dominant, 0), "avg_launch_us": avg_us[dominant]}
        head = profile_head()
        result["roofline"]["profile_head"] = head  # (what `traffic` and the counter-based figures below were measured on)
        if head.get("matches_running_library") is False:
            print("bench: the committed counter summaries (" + str(head.get("profile")) + ") were taken on library build " + str(head.get("library_build_id")) +
                  ", this run is build " + str(head.get("running_library_build_id")) + ": traffic / issue bounds are those of the older kernels", file=sys.stderr)
        if eval_traffic:  # what really crosses the HBM interface, by the PMC counters of the committed profile
            result["eval_hbm_fraction_counter_bytes"] = (eval_traffic / (ms_per_step * 1e-3)) / (HBM_PEAK_GBS * 1e9)
            if traffic:
                result["roofline"]["frac_counter_bytes"] = traffic / (avg_us[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS
        # The roofs that actually bind the pair kernels: vector-instruction issue (every VALU instruction of a wave holds
        # its SIMD for 4 cycles at FP64 rate).  Instruction counts: SQ_INSTS_VALU of the committed counter pass where there
        # is one for this workload, else the hand-read table x the wave-steps this geometry makes the kernels execute.
        counters, counter_file = counter_valu_instructions(args.system, mode)
        issue = []
        steps_by_kernel = pair_wave_steps(system, rep.geoms[W]) if any(k in PAIR_STEP_VALU for k in avg_us) else {}
        for kname in ISSUE_BOUND_KERNELS:
            if kname not in avg_us:
                continue
            if counters and kname in counters:
                valu, source = counters[kname]["valu"], counter_file + ": SQ_INSTS_VALU"
            elif kname in PAIR_STEP_VALU:
                valu = sum(steps_by_kernel[kname][kind] * v for kind, (v, _) in PAIR_STEP_VALU[kname].items())
                source = "hand-read instruction table x wave-steps of this geometry"
            else:
                continue
            bound_us = valu * 4 / SIMDS / (CLOCK_GHZ * 1e3)
            entry = {"bound": "fp64_issue", "kernel": kname, "valu_instructions": int(valu), "source": source, "cycles_per_instruction": 4,
                     "bound_us": round(bound_us, 2), "avg_launch_us": round(avg_us[kname], 2),
                     "frac": round(bound_us / avg_us[kname], 3) if avg_us[kname] > 0 else None}
            if counters and kname in counters:
                entry["lds_bank_conflict_share"] = counters[kname]["lds_conflict_share"]
            issue.append(entry)
        for entry in issue:
            entry["profile_head"] = head.get("git_head") or head.get("library_build_id")
        result["rooflines_issue"] = issue
        # speed of light of the evaluation as launched: per kernel the roof that binds it -- vector issue for the pair
        # kernels, the modelled bytes at the HBM peak for the rest (tree kernels: latency-bound far above that)
        issue_by_kernel = {e["kernel"]: e["bound_us"] for e in issue}
        sol = {k: issue_by_kernel.get(k, b_kernel.get(k, 0) / (HBM_PEAK_GBS * 1e9) * 1e6) for k in avg_us}
        result["speed_of_light_us"] = {"sum": round(sum(sol.values()), 2), "per_kernel": {k: round(v, 2) for k, v in sol.items()},
                                       "frac_of_measured": round(sum(sol.values()) / (1e3 * ms_per_step), 3),
                                       "profile_head": head.get("git_head") or head.get("library_build_id"),
                                       "profile_matches_running_library": head.get("matches_running_library")}
        result["kernel_avg_us"] = {k: round(v, 2) for k, v in avg_us.items()}
        result["kernel_avg_us_note"] = ("hipEvent intervals of a second pass minus one uniform event overhead chosen so that the kernels sum to the "
                                        "measured step (they do by construction: this is a SPLIT of ms_per_step, not a check of it; the independent "
                                        "per-kernel durations are the rocprofv3 summary under profiles/)")
        result["launches_per_evaluation"] = len(avg_us)
        result["kernel_event_us"] = {k: round(v, 2) for k, v in raw_us.items()}
        result["event_overhead_us"] = round(event_overhead_us, 2)

        return {}

    def cpu_baseline_section():
        # ---- CPU baseline (rank 0, on a bounded sample of the timed geometries; a smaller one when other replicas ran too)
        if args.cpu_evals > 0:
            evals = min(args.cpu_evals if world == 1 else min(args.cpu_evals, 5), K)
            oracle_kw = {"cutoff": 1.0} if mode in ("fast", "fast+single") else {}
            cpu_ms, de, df = cpu_baseline_leg(system, rep.geoms[W:], rep.host_results(W, evals), evals, **oracle_kw)
            result["cpu_baseline"] = {"value": 86.4 / cpu_ms, "unit": "ns/day", "ms_per_eval": cpu_ms, "cores": 1, "kind": "port",
                                      "sample": f"first {evals} of the {K} timed geometries of rank 0, single-threaded FP64 oracle (oracle/agbnp_oracle.cpp, g++ -O2)"}
            result["parity_on_sample"] = {"max_abs_dE_kJmol": de, "max_abs_dF_kJmolnm": df,
                                          "tolerance": "single-precision pair terms: no parity bar" if mode == "fast+single" else 1e-4}

        return {}

    # ---- BASELINE.json's other configurations, bounded (one GPU, Reference-semantics run only).  Every record below is
    #      OPTIONAL: it runs under `line.optional`, which turns a failure into {"error": ...} inside its own entry (and a
    #      line on stderr), and under the line's watchdog, which prints the line with what it has when the records overrun
    #      their time budget -- the contract's line never depends on them (BENCH_r05: a secondary that would not settle took
    #      the whole line with it).
    def optional_records():
        nonlocal rep
        del rep
        # what the timed region above never contains: a rebuild evaluation, and the amortised cost of
        # rebuilds and re-plans along a walk that drifts
        rb = line.optional("neighbour_rows.rebuild", rebuild_entry, dev, "1dwc", 200, 20)
        if "neighbour_rows" in result and "rebuild_eval_ms" in rb:
            line.update("neighbour_rows", {"rebuild_eval_ms": rb["rebuild_eval_ms"], "plain_eval_ms_same_box": rb["plain_eval_ms"],
                                           "rebuild_cost_ms": rb["rebuild_cost_ms"], "builds_in_rebuild_timed_region": rb["builds_in_rebuild_timed_region"],
                                           "builds_in_headline_timed_region": rb["builds_in_plain_timed_region"], "rebuild_note": rb["note"]})
        elif "error" in rb:
            line.update("neighbour_rows", {"rebuild_error": rb})
        line.put("drift", line.optional("drift", drift_entry, dev, "1dwc", args.drift_steps, args.drift_sigma))
        line.put("secondary", [])
        for config, a, kw in (
                ("1: trpcage GaussVol (version 0), NoCutoff", ("trpcage", 0, 200, 20, 20), dict(method=P.AGBNPForce.NoCutoff)),
                ("2: trpcage AGBNP1 (version 1), CutoffNonPeriodic 1.2 nm", ("trpcage", 1, 200, 20, 20), dict(cutoff=1.2)),
                ("4: HIV-RT stand-in = 2x2x1 lattice of 1dwc (synthetic), AGBNP1", ("1dwc_x4", 1, 40, 6, 1), {}),
                ("(bundled example, not a BASELINE.json config) 2clr, 5983 atoms, AGBNP1: example/2clr_agbnp1.dms", ("2clr", 1, 200, 20, 2), {})):
            line.append("secondary", dict(config=config, **line.optional("secondary: " + config, secondary_entry, dev, *a, **kw)))
        # the other evaluation modes on the headline workload (each has a line of its own with --mode; here for the record)
        line.put("other_modes", [])
        for m in ("fast", "fast+single", "deterministic"):
            line.append("other_modes", dict(mode=m, **line.optional("other_modes: " + m, secondary_entry, dev, "1dwc", 1, 200, 20, 0, cutoff=1.0, mode=m)))
        # the headline's own configuration with the preparation launch of rounds 1-4 (six launches), same box, same protocol:
        # what the five-launch mode is worth in THIS run
        line.append("other_modes", dict(mode="reference, six launches (AGBNP_HIP_FIVE_LAUNCHES=0)",
                                        **line.optional("other_modes: six launches", _with_env, {"AGBNP_HIP_FIVE_LAUNCHES": "0"},
                                                        lambda: secondary_entry(dev, "1dwc", 1, 200, 20, 0))))
        line.put("concurrent_replicas_on_one_gpu", [line.optional(f"concurrent replicas x{r}", concurrent_replicas_entry, dev, "1dwc", r, 200, 20) for r in (2, 4)])
        line.put("openmm_entry", line.optional("openmm_entry", openmm_entry, dev, "1dwc", 200, 20))
        line.put("openmm_entry_particle_order", line.optional("openmm_entry_particle_order", openmm_entry, dev, "1dwc", 200, 20, shuffled=False))
        line.put("md_loop", line.optional("md_loop", md_loop_entry, dev, "1dwc", 3000))

    line = Line(result)
    try:
        # `roofline` and `cpu_baseline` belong to the contract: a failure there is reported in the line (and on stderr) and
        # the line is still printed -- a line without one of them says why
        r = line.optional("roofline", roofline_section)
        if "error" in r:
            line.put("roofline_error", r)
        r = line.optional("cpu_baseline", cpu_baseline_section)
        if "error" in r:
            line.put("cpu_baseline_error", r)
        if world == 1 and args.secondary and mode is None and args.system == "1dwc":
            line.watchdog(args.secondary_budget_s)
            optional_records()
    finally:
        line.emit()
    return 0


def launch_replicas(args, argv):
    """`python3 bench.py --gpus N` with no launcher around it: THIS process -- which has imported neither torch nor the
    engine and never touches a GPU -- starts N fresh python processes, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them; every rank takes device LOCAL_RANK), relays rank 0's
    standard output (the JSON line), sends the other ranks' to standard error, and returns the worst exit code.  A rank that
    dies takes the job with it: the others get a grace period to leave by themselves (they do, through Job.abort, if the
    dead rank reached a collective) and are then terminated.  Nothing that has initialised HIP is ever exec'ed: the
    children are new processes started from a parent that never loaded the runtime."""
    import signal
    import socket
    import subprocess
    import threading

    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                 "AGBNP_BENCH_LAUNCHER": "bench.py self-launch (subprocess per rank)"})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (this pool's driver only supports dmabuf IPC: RCCL needs it)
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def relay(pipe):  # rank 0's JSON line goes to standard output; whatever else a library prints there, to standard error
        for line in iter(pipe.readline, b""):
            out = sys.stdout.buffer if line.lstrip().startswith(b"{") else sys.stderr.buffer
            out.write(line)
            out.flush()

    relay_thread = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    relay_thread.start()

    def stop(sig, _frame):  # the driver's timeout / Ctrl-C: the ranks go with the parent
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        raise SystemExit(128 + sig)

    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)

    grace, failed_at = float(os.environ.get("AGBNP_BENCH_GRACE_SECONDS", "30")), None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            print(f"bench: rank(s) {bad} failed; the others have {grace:.0f} s to leave", file=sys.stderr, flush=True)
        if failed_at is not None and time.monotonic() - failed_at > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    relay_thread.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    worst = max((c if c >= 0 else 128 - c) for c in codes)  # (a rank killed by signal s counts as 128 + s)
    if worst:
        print(f"bench: exit codes by rank {codes}", file=sys.stderr, flush=True)
    return worst


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--system", default="1dwc")
    ap.add_argument("--cpu-evals", type=int, default=20, help="size of the CPU-baseline sample (0 disables the leg; at most 5 when N > 1)")
    ap.add_argument("--secondary", type=int, default=1, help="also time BASELINE.json's other configurations (one GPU only)")
    ap.add_argument("--secondary-budget-s", type=float, default=300.0,
                    help="time budget of the optional records behind the headline; when it runs out the line is printed with what it has")
    ap.add_argument("--mode", default="reference", choices=["reference", "fast", "fast+single", "deterministic"],
                    help="fast = the OpenCL platform's semantics (cutoff on every pair stage); deterministic = bit-reproducible "
                         "sums (Reference semantics); each printed as its own line")
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed evaluations before the timed region until the device has been busy this long (clock ramp)")
    ap.add_argument("--replicas-per-gpu", type=int, default=1,
                    help="also report the aggregate of R concurrent replicas per GPU (streams of one process); the headline stays 1")
    ap.add_argument("--drift-steps", type=int, default=2000, help="evaluations of the random-walk (drift) record")
    ap.add_argument("--drift-sigma", type=float, default=0.001, help="nm per coordinate and step of that walk")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_replicas(args, argv)  # no launcher around this process: it becomes the launcher (and only that)
    if args.gpus > 1 and env_world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} inside a launcher needs WORLD_SIZE={args.gpus} (torch.distributed.run --nproc-per-node {args.gpus}); WORLD_SIZE={env_world}")
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
