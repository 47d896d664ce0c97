"""Device-resident molecular dynamics around the AGBNP engine, for the py3 counterparts of the reference's example
scripts (example/test_agbnp.py: minimise, Langevin equilibration, NVE energy-conservation run; example/1dwc_benchmark.py:
Langevin timing run) and for the energy-conservation test.

The reference gets its bonded and Coulomb/LJ terms from OpenMM's OPLS system (DesmondDMSFile.createSystem), which is
outside this repository; here the only force-field term besides AGBNP is a harmonic tether of every atom to its start
position, which keeps the geometry a protein.  Everything lives on the GPU (torch tensors for the integrator state,
`agbnp_hip_execute_device` for the force); one MD step is captured ONCE as a HIP graph and replayed, the host only
synchronises every `check_every` steps to read the engine's overflow log (agbnp_hip_finish).

PyTorch is plumbing here (device arrays, the graph capture API, normal random numbers), not the product.
"""
import numpy as np

KB = 0.0083144626  # kJ/mol/K


class DeviceMD:
    def __init__(self, system, kernel, k_tether=2.0e4, dt=0.001, temperature=300.0, friction=1.0, seed=0, device="cuda:0",
                 log_capacity=200000):
        import torch
        self.torch = torch
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
        self.ene = torch.zeros(1, **f64)   # potential energy of the last force evaluation (tethers + AGBNP)
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
        self.graphs = {}
        self.generation = None
        self.steps_done = 0

    # ---- force field: tethers + AGBNP (added on the device by the engine)
    def forces(self):
        torch = self.torch
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

    def step_verlet(self):  # velocity Verlet (the reference's NVE check uses OpenMM's VerletIntegrator, test_agbnp.py:57)
        self.v.addcmul_(self.frc, self.hdt_m)
        self.x.add_(self.v, alpha=self.dt)
        self.forces()
        self.v.addcmul_(self.frc, self.hdt_m)
        self._record()

    def step_langevin(self):  # BAOAB (the reference uses LangevinIntegrator(300 K, 1/ps), test_agbnp.py:37, 1dwc_benchmark.py:20)
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

    def _graph(self, kind):
        torch = self.torch
        if self.generation != self.kernel.generation():  # first use, or the capacity variant was raised: kernels are stale
            self.graphs.clear()
            self.generation = self.kernel.generation()
        if kind not in self.graphs:
            step = {"verlet": self.step_verlet, "langevin": self.step_langevin, "descent": self.step_descent}[kind]
            counter0 = self.counter.clone()
            state = (self.x.clone(), self.v.clone(), self.frc.clone(), self.ene.clone())
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):  # torch wants a few eager runs on a side stream before a capture
                step()
                self.kernel.finish(side.cuda_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            torch.cuda.synchronize()
            # the capture itself does not run the step, the warm-up did: put the state back
            for dst, src in zip((self.x, self.v, self.frc, self.ene), state):
                dst.copy_(src)
            self.counter.copy_(counter0)
            self.graphs[kind] = g
        return self.graphs[kind]

    def run(self, nsteps, kind="langevin", check_every=1000, on_report=None):
        """Replays the captured step; every `check_every` steps synchronises and reads the engine's overflow log.  Returns
        the number of steps whose AGBNP contribution was withheld (tree capacity exceeded): 0 in a healthy run."""
        torch = self.torch
        missed = 0
        done = 0
        while done < nsteps:
            chunk = min(check_every, nsteps - done)
            g = self._graph(kind)
            for _ in range(chunk):
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

