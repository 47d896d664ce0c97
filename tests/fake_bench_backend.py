"""CPU double of bench.HipBackend for the multi-process tests of bench.py (AGBNP_BENCH_BACKEND_MODULE=tests.fake_bench_backend):
no GPU, no engine -- the worker's collectives, retry / abort decisions, self-launcher and JSON line run as they do on GPUs.
The fake kernel's host-facing execute() answers with the CPU oracle (this is test infrastructure), so the cpu_baseline leg
of the line is exercised too.

Scripted through the environment, per rank r:
  AGBNP_FAKE_WITHHELD_<r>   comma-separated withheld counts that finish() plays back (then 0 for ever)
  AGBNP_FAKE_RAISE_<r>      execute_device raises at its k-th call (a HIP error on one rank alone)
"""
import os
import time

import numpy as np


class FakeKernel:
    NAMES = ("k_prep", "k_tree_cavity", "k_born_rows", "k_gb_tiles", "k_dborn_rows", "k_tree_pseudo")

    def __init__(self, rank):
        self.script = [int(v) for v in os.environ.get(f"AGBNP_FAKE_WITHHELD_{rank}", "").split(",") if v.strip()]
        self.raise_at = int(os.environ.get(f"AGBNP_FAKE_RAISE_{rank}", "0"))
        self.calls = self.finishes = 0
        self.profiling = False
        self.profiled = 0

    def initialize(self, force):
        from oracle import Oracle
        self.force = force
        self.oracle = Oracle(*force._arrays(), version=force.getVersion())
        self.n = force.getNumParticles()

    def execute_device(self, d_pos, d_force, d_energy, stream=None):
        self.calls += 1
        if self.raise_at and self.calls == self.raise_at:
            raise RuntimeError("fake HIP error on this rank")
        if self.profiling:
            self.profiled += 1
        time.sleep(2e-4)

    def finish(self, stream=None):
        self.finishes += 1
        return self.script.pop(0) if self.script else 0

    def execute(self, positions, forces):
        e, f = self.oracle.execute(np.asarray(positions))
        forces += f
        return e

    def scalar(self, name):
        return {"total_nodes": 12732.0, "variant": 0.0, "rows_on": 1.0, "row_builds": 2.0, "row_slice": 256.0, "pack_plans": 5.0, "forests": 144.0, "pack_level": 0.0}[name]

    def set_profiling(self, enabled):
        self.profiling, self.profiled = bool(enabled), 0

    def kernel_times(self):
        return {k: (0.01 * (i + 1) * self.profiled, self.profiled) for i, k in enumerate(self.NAMES)}


class Backend:
    name = "fake"

    def __init__(self, torch, local_rank, collective_backend):
        self.torch, self.index, self.device = torch, local_rank, torch.device("cpu")

    def synchronize(self):
        pass

    def wait_idle(self):
        pass

    def current_stream(self):
        return 0

    def tensor(self, array, dtype):
        return self.torch.tensor(np.asarray(array), dtype=dtype).contiguous()

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=dtype)

    def identity(self):
        return {"device_index": self.index, "device_name": "fake device", "device_uuid": f"fake-{self.index}", "pci_bus_id": self.index}

    def kernel(self, mode=None):
        return FakeKernel(int(os.environ.get("RANK", "0")))
