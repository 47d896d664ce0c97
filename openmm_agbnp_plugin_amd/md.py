"""Device-resident molecular dynamics around the AGBNP engine, for the py3 counterparts of the reference's example
scripts (example/test_agbnp.py: minimise, Langevin equilibration, NVE energy-conservation run; example/1dwc_benchmark.py:
Langevin timing run) and for the energy-conservation test.

The reference gets its bonded and Coulomb/LJ terms from OpenMM's OPLS system (DesmondDMSFile.createSystem), which is
outside this repository; here the only force-field term besides AGBNP is a harmonic tether of every atom to its start
position, which keeps the geometry a protein.  Everything lives on the GPU (torch tensors for the integrator state,
`agbnp_hip_execute_device` for the force); one MD step is captured ONCE as a HIP graph and replayed, the host only
synchronises every `check_every` steps to read the engine's overflow log (agbnp_hip_finish).

The integrator itself is two launches of libagbnp_md.so (csrc/md_kernels.hip: everything in front of the force
evaluation, everything behind it -- between the steps of a run both in ONE launch; Philox normal deviates) around the six
of the AGBNP evaluation; written in torch operations it is seventeen (`fused=False`, kept as the cross-check of the
kernels): 0.163 -> 0.11 ms per step of 1dwc.

PyTorch is plumbing here (device arrays, the graph capture API), not the product.
"""
import ctypes as C
import os

import numpy as np

KB = 0.0083144626  # kJ/mol/K

_MD_LIB = None


def _md_lib():
    """libagbnp_md.so (example support, not part of the drop-in boundary); built by csrc/Makefile next to the engine."""
    global _MD_LIB
    if _MD_LIB is None:
        from . import _lib
        _lib.load()  # (one HIP runtime per process: the engine's loader settles which)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libagbnp_md.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path)
        vp, dbl = C.c_void_p, C.c_double
        lib.agbnp_md_blocks.argtypes = [C.c_int]
        lib.agbnp_md_pre.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, C.c_ulonglong, vp, vp, vp]
        lib.agbnp_md_post.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_longlong, vp, vp]
        lib.agbnp_md_tethers.argtypes = [C.c_int, vp, vp, vp, dbl, vp, vp]
        lib.agbnp_md_mid.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, C.c_ulonglong, vp, vp, vp, vp, vp, vp, vp, vp,
                                     C.c_longlong, vp, vp]
        _MD_LIB = lib
    return _MD_LIB


class DeviceMD:
    def __init__(self, system, kernel, k_tether=2.0e4, dt=0.001, temperature=300.0, friction=1.0, seed=0, device="cuda:0",
                 log_capacity=200000, fused=True):
        import torch
        self.torch = torch
        self.fused = bool(fused)
        self.seed = int(seed)
        self.system, self.kernel = system, kernel
        self.dev = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.dt, self.k = float(dt), float(k_tether)
        self.T, self.gamma = float(temperature), float(friction)
        # masses in amu: hydrogens 1.008, heavy atoms carbon-like (the .dat fixtures carry no element)
        self.mass = torch.tensor(np.where(system.ishydrogen == 1, 1.008, 12.0)[:, None], **f64)
        self.x0 = torch.tensor(system.pos, **f64)
        self.x = self.x0.clone()
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        self.gen = gen
        self.v = torch.randn(self.x.shape, generator=gen, **f64) * torch.sqrt(KB * self.T / self.mass)
        self.frc = torch.zeros_like(self.x)
        self.last = torch.zeros(2, **f64)  # {potential, kinetic} energy of the last step
        self.ene = self.last[0:1]          # potential energy of the last force evaluation (tethers + AGBNP)
        self.noise = torch.empty_like(self.x)
        self.c1 = float(np.exp(-self.gamma * self.dt))
        self.c2 = torch.sqrt((1.0 - self.c1 * self.c1) * KB * self.T / self.mass)
        # per-step log written inside the graph: potential and kinetic energy at index `counter`
        self.log_pe = torch.zeros(log_capacity, **f64)
        self.log_ke = torch.zeros(log_capacity, **f64)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.one = torch.ones(1, dtype=torch.int64, device=self.dev)
        # Every torch op below is one tiny launch (~2 us each inside the replayed graph), so the step is written with as
        # few of them as the arithmetic allows: fused multiply-adds, preallocated outputs, dot products for the sums.
        self.hdt_m = (0.5 * self.dt) / self.mass  # dt / 2m
        self.d = torch.zeros_like(self.x)           # x - x0
        self.mv = torch.zeros_like(self.x)          # m v
        self.ke = torch.zeros(1, **f64)
        # fused integrator (csrc/md_kernels.hip): the word the engine adds the AGBNP energy to (handed back as zero by every
        # step), the tether energy as per-block partials, the kinetic-energy accumulator and the arrival counter
        self.n = int(system.n)
        self.log_capacity = int(log_capacity)
        if self.fused:
            lib = _md_lib()
            self.e_agbnp = torch.zeros(1, **f64)
            self.tether_part = torch.zeros(lib.agbnp_md_blocks(self.n), **f64)
            self.tether_part2 = torch.zeros_like(self.tether_part)  # (a run of steps alternates between the two: k_md_mid)
            self.acc = torch.zeros(2, **f64)
            self.done = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self.hdt_m1 = self.hdt_m.reshape(-1).contiguous()
            self.c2_1 = self.c2.reshape(-1).contiguous()
            self.mass1 = self.mass.reshape(-1).contiguous()
        self._eager = None
        self.graphs = {}
        self.generation = None
        self.steps_done = 0

    # ---- force field: tethers + AGBNP (added on the device by the engine)
    def forces(self):
        """Tethers + AGBNP at the current positions: self.frc, self.ene.  Inside a graph capture it joins the capture; called
        eagerly on torch's default (null) stream it runs on a stream of its own and waits for it -- the engine takes a null
        stream for its context's own stream, which nothing of torch's is ordered against."""
        torch = self.torch
        if not torch.cuda.is_current_stream_capturing() and torch.cuda.current_stream().cuda_stream == 0:
            if self._eager is None:
                self._eager = torch.cuda.Stream(device=self.dev)
            self._eager.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._eager):
                self._forces()
            self._eager.synchronize()
            return
        self._forces()

    def _forces(self):
        torch = self.torch
        if self.fused:
            st = torch.cuda.current_stream().cuda_stream
            self.e_agbnp.zero_()
            self._check(_md_lib().agbnp_md_tethers(self.n, self.x.data_ptr(), self.x0.data_ptr(), self.frc.data_ptr(), self.k, self.tether_part.data_ptr(), st))
            self.kernel.execute_device(self.x.data_ptr(), self.frc.data_ptr(), self.e_agbnp.data_ptr(), st)
            torch.add(self.tether_part.sum().reshape(1), self.e_agbnp, out=self.ene)
            self.e_agbnp.zero_()  # (a step that follows starts its own sum)
            return
        torch.sub(self.x, self.x0, out=self.d)
        torch.mul(self.d, -self.k, out=self.frc)
        torch.mul(torch.dot(self.d.view(-1), self.d.view(-1)).reshape(1), 0.5 * self.k, out=self.ene)
        self.kernel.execute_device(self.x.data_ptr(), self.frc.data_ptr(), self.ene.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def _record(self):
        torch = self.torch
        torch.mul(self.v, self.mass, out=self.mv)
        torch.mul(torch.dot(self.mv.view(-1), self.v.view(-1)).reshape(1), 0.5, out=self.ke)
        self.log_pe.index_copy_(0, self.counter, self.ene)
        self.log_ke.index_copy_(0, self.counter, self.ke)
        self.counter.add_(self.one)

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise RuntimeError(f"libagbnp_md.so: launch failed (hipError {rc})")

    def _step_fused(self, kind):
        """One step in two launches around the AGBNP evaluation (csrc/md_kernels.hip)."""
        lib, st = _md_lib(), self.torch.cuda.current_stream().cuda_stream
        self._check(lib.agbnp_md_pre(self.n, kind, self.x.data_ptr(), self.v.data_ptr(), self.frc.data_ptr(), self.x0.data_ptr(),
                                     self.hdt_m1.data_ptr(), self.c2_1.data_ptr(), self.c1, self.dt, self.k, self.seed, self.counter.data_ptr(),
                                     self.tether_part.data_ptr(), st))
        self.kernel.execute_device(self.x.data_ptr(), self.frc.data_ptr(), self.e_agbnp.data_ptr(), st)
        self._check(lib.agbnp_md_post(self.n, self.v.data_ptr(), self.frc.data_ptr(), self.hdt_m1.data_ptr(), self.mass1.data_ptr(),
                                      self.e_agbnp.data_ptr(), self.tether_part.data_ptr(), self.acc.data_ptr(), self.done.data_ptr(),
                                      self.log_pe.data_ptr(), self.log_ke.data_ptr(), self.counter.data_ptr(), self.log_capacity,
                                      self.last.data_ptr(), st))

    def _steps_fused(self, kind, steps):
        """`steps` consecutive steps: front half, then (evaluation, back half + next front half in ONE launch) between the
        steps, evaluation, back half: 7 launches per step instead of 8."""
        lib, st = _md_lib(), self.torch.cuda.current_stream().cuda_stream
        parts = (self.tether_part, self.tether_part2)
        self._check(lib.agbnp_md_pre(self.n, kind, self.x.data_ptr(), self.v.data_ptr(), self.frc.data_ptr(), self.x0.data_ptr(),
                                     self.hdt_m1.data_ptr(), self.c2_1.data_ptr(), self.c1, self.dt, self.k, self.seed, self.counter.data_ptr(),
                                     parts[0].data_ptr(), st))
        for j in range(steps):
            self.kernel.execute_device(self.x.data_ptr(), self.frc.data_ptr(), self.e_agbnp.data_ptr(), st)
            old = parts[j % 2]
            if j + 1 < steps:
                self._check(lib.agbnp_md_mid(self.n, kind, self.x.data_ptr(), self.v.data_ptr(), self.frc.data_ptr(), self.x0.data_ptr(),
                                             self.hdt_m1.data_ptr(), self.mass1.data_ptr(), self.c2_1.data_ptr(), self.c1, self.dt, self.k, self.seed,
                                             self.e_agbnp.data_ptr(), old.data_ptr(), parts[(j + 1) % 2].data_ptr(), self.acc.data_ptr(),
                                             self.done.data_ptr(), self.log_pe.data_ptr(), self.log_ke.data_ptr(), self.counter.data_ptr(),
                                             self.log_capacity, self.last.data_ptr(), st))
            else:
                self._check(lib.agbnp_md_post(self.n, self.v.data_ptr(), self.frc.data_ptr(), self.hdt_m1.data_ptr(), self.mass1.data_ptr(),
                                              self.e_agbnp.data_ptr(), old.data_ptr(), self.acc.data_ptr(), self.done.data_ptr(),
                                              self.log_pe.data_ptr(), self.log_ke.data_ptr(), self.counter.data_ptr(), self.log_capacity,
                                              self.last.data_ptr(), st))

    def step_verlet(self):  # velocity Verlet (the reference's NVE check uses OpenMM's VerletIntegrator, test_agbnp.py:57)
        if self.fused:
            return self._step_fused(1)
        self.v.addcmul_(self.frc, self.hdt_m)
        self.x.add_(self.v, alpha=self.dt)
        self.forces()
        self.v.addcmul_(self.frc, self.hdt_m)
        self._record()

    def step_langevin(self):  # BAOAB (the reference uses LangevinIntegrator(300 K, 1/ps), test_agbnp.py:37, 1dwc_benchmark.py:20)
        if self.fused:
            return self._step_fused(0)
        self.v.addcmul_(self.frc, self.hdt_m)
        self.x.add_(self.v, alpha=0.5 * self.dt)
        self.noise.normal_(generator=None)
        self.v.mul_(self.c1).addcmul_(self.c2, self.noise)
        self.x.add_(self.v, alpha=0.5 * self.dt)
        self.forces()
        self.v.addcmul_(self.frc, self.hdt_m)
        self._record()

    def step_descent(self, gain=2.0e-6):  # crude minimiser: a capped move along the force
        move = (gain * self.frc).clamp_(-0.002, 0.002)
        self.x.add_(move)
        self.forces()
        self._record()

    # ---- graph capture / replay
    def settle(self):
        """Outside any capture: first evaluations (allocations, capacity negotiation, forest packing)."""
        torch = self.torch
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(8):
                for _ in range(3):
                    self.forces()
                if self.kernel.finish(side.cuda_stream) == 0:
                    break
            else:
                raise RuntimeError("AGBNP capacity negotiation did not converge")
        torch.cuda.synchronize()

    def _graph(self, kind, steps=1):
        """A HIP graph of `steps` consecutive MD steps of the given kind (captured once per kind, length and engine generation)."""
        torch = self.torch
        if self.generation != self.kernel.generation():  # first use, or the capacity variant was raised: kernels are stale
            self.graphs.clear()
            self.generation = self.kernel.generation()
        if (kind, steps) not in self.graphs:
            step = {"verlet": self.step_verlet, "langevin": self.step_langevin, "descent": self.step_descent}[kind]
            counter0 = self.counter.clone()
            state = (self.x.clone(), self.v.clone(), self.frc.clone(), self.last.clone())
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):  # torch wants a few eager runs on a side stream before a capture
                step()
                self.kernel.finish(side.cuda_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if self.fused and steps > 1 and kind in ("langevin", "verlet"):
                    self._steps_fused(0 if kind == "langevin" else 1, steps)
                else:
                    for _ in range(steps):
                        step()
            torch.cuda.synchronize()
            # the capture itself does not run the steps, the warm-up did: put the state back
            for dst, src in zip((self.x, self.v, self.frc, self.last), state):
                dst.copy_(src)
            self.counter.copy_(counter0)
            self.graphs[(kind, steps)] = g
        return self.graphs[(kind, steps)]

    def run(self, nsteps, kind="langevin", check_every=1000, on_report=None, steps_per_graph=10):
        """Replays the captured steps (`steps_per_graph` MD steps per graph launch: the host's share of a launch is paid once
        for all of them); every `check_every` steps synchronises and reads the engine's overflow log.  Returns the number of
        steps whose AGBNP contribution was withheld (tree capacity exceeded): 0 in a healthy run."""
        torch = self.torch
        missed = 0
        done = 0
        while done < nsteps:
            chunk = min(check_every, nsteps - done)
            many = chunk // steps_per_graph if steps_per_graph > 1 else 0
            if many:
                g = self._graph(kind, steps_per_graph)
                for _ in range(many):
                    g.replay()
            rest = chunk - many * steps_per_graph
            if rest:
                g = self._graph(kind)
                for _ in range(rest):
                    g.replay()
            done += chunk
            self.steps_done += chunk
            missed += self.kernel.finish(torch.cuda.current_stream().cuda_stream)
            if on_report:
                on_report(self)
        return missed

    # ---- observables
    def energies(self, last=None):
        """(potential, kinetic) per recorded step as numpy arrays."""
        n = int(self.counter.item())
        pe, ke = self.log_pe[:n].cpu().numpy(), self.log_ke[:n].cpu().numpy()
        if last:
            pe, ke = pe[-last:], ke[-last:]
        return pe, ke

    def temperature(self):
        ke = 0.5 * float((self.mass * self.v * self.v).sum())
        return 2.0 * ke / (3 * self.system.n * KB)

