// Integrator steps for the device-resident MD driver of the example scripts (openmm_agbnp_plugin_amd/md.py; the py3
// counterparts of the reference's example/1dwc_benchmark.py and example/test_agbnp.py).  NOT part of the drop-in boundary
// (include/agbnp_hip.h): in the reference the integrator is OpenMM's.  A step written in torch operations is seventeen tiny
// launches around the six of the AGBNP evaluation (0.163 ms per step of 1dwc against 0.104 for the evaluation alone); here
// it is two: everything in front of the force evaluation, everything behind it.
//
//   pre   Langevin (BAOAB, the reference's LangevinIntegrator(300 K, 1/ps, 1 fs), 1dwc_benchmark.py:20):
//           v += dt/2m f;  x += dt/2 v;  v = c1 v + c2 xi;  x += dt/2 v
//         velocity Verlet (the reference's NVE check, test_agbnp.py:57):   v += dt/2m f;  x += dt v
//         then the tethers, the only force-field term besides AGBNP:  f = -k (x - x0), their energy as per-block partials
//   post  v += dt/2m f (f now holds tethers + AGBNP);  kinetic energy;  potential = tether partials + what the engine added
//         to the energy word;  both go into the per-step logs;  the energy word and the accumulators are handed back as zeros
//
// Normal deviates: Philox4x32-10 keyed by the seed, counter = (step number on the device, atom), Box-Muller in FP64 on
// 53-bit uniforms: the stream of a run depends on nothing but the seed (graph replays included: the step number is read
// from device memory).
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int kBlock = 256;

struct Philox {
  uint32_t c[4];
};
__device__ __forceinline__ Philox philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
    k0 += W0, k1 += W1;
  }
  return Philox{{c0, c1, c2, c3}};
}
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b) {  // (0, 1]
  const uint64_t u = ((uint64_t)a << 21) ^ (uint64_t)(b >> 11);  // 53 bits
  return ((double)(u & ((1ull << 53) - 1ull)) + 1.0) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ void box_muller(double u1, double u2, double& z0, double& z1) {
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z0 = r * c, z1 = r * s;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < kBlock / 64; w++) r += red[w];  // fixed order
  __syncthreads();
  return r;
}

// the front half of a step for atom i, velocity pv already kicked: drift (+ OU for Langevin), tethers; returns the atom's
// tether energy.  kind 0: Langevin (BAOAB), 1: velocity Verlet
__device__ __forceinline__ double front_half(int i, int kind, double (&px)[3], double (&pv)[3], double* __restrict__ x, double* __restrict__ v,
                                             double* __restrict__ f, const double* __restrict__ x0, const double* __restrict__ c2, double c1,
                                             double dt, double ktether, unsigned long long seed, unsigned long long s) {
  if (kind == 0) {
    const Philox a = philox4x32((uint32_t)i, (uint32_t)s, (uint32_t)(s >> 32), 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const Philox b = philox4x32((uint32_t)i, (uint32_t)s, (uint32_t)(s >> 32), 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
    double z[4];
    box_muller(uniform53(a.c[0], a.c[1]), uniform53(a.c[2], a.c[3]), z[0], z[1]);
    box_muller(uniform53(b.c[0], b.c[1]), uniform53(b.c[2], b.c[3]), z[2], z[3]);
    const double cn = c2[i];
    for (int d = 0; d < 3; d++) {
      px[d] = fma(0.5 * dt, pv[d], px[d]);
      pv[d] = fma(c1, pv[d], cn * z[d]);
      px[d] = fma(0.5 * dt, pv[d], px[d]);
    }
  } else {
    for (int d = 0; d < 3; d++) px[d] = fma(dt, pv[d], px[d]);
  }
  double e = 0.0;
  for (int d = 0; d < 3; d++) {
    const double dd = px[d] - x0[3 * i + d];
    x[3 * i + d] = px[d];
    v[3 * i + d] = pv[d];
    f[3 * i + d] = -ktether * dd;
    e = fma(0.5 * ktether * dd, dd, e);
  }
  return e;
}

// one thread per atom: everything in front of the force evaluation of a step
__global__ __launch_bounds__(kBlock) void k_md_pre(int n, int kind, double* __restrict__ x, double* __restrict__ v, double* __restrict__ f,
                                                  const double* __restrict__ x0, const double* __restrict__ hdt_m,
                                                  const double* __restrict__ c2, double c1, double dt, double ktether,
                                                  unsigned long long seed, const long long* __restrict__ step,
                                                  double* __restrict__ tether_part) {
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  double e = 0.0;
  if (i < n) {
    const double h = hdt_m[i];
    double px[3], pv[3];
    for (int d = 0; d < 3; d++) px[d] = x[3 * i + d], pv[d] = fma(h, f[3 * i + d], v[3 * i + d]);
    e = front_half(i, kind, px, pv, x, v, f, x0, c2, c1, dt, ktether, seed, (unsigned long long)step[0]);
  }
  e = block_sum(e, red);
  if (threadIdx.x == 0) tether_part[blockIdx.x] = e;
}

// acc: {kinetic energy sum, -} (doubles), done: blocks that have added theirs.  The last block to arrive writes the logs.
__global__ __launch_bounds__(kBlock) void k_md_post(int n, double* __restrict__ v, const double* __restrict__ f, const double* __restrict__ hdt_m,
                                                   const double* __restrict__ mass, double* __restrict__ energy,
                                                   const double* __restrict__ tether_part, double* __restrict__ acc,
                                                   unsigned* __restrict__ done, double* __restrict__ log_pe, double* __restrict__ log_ke,
                                                   long long* __restrict__ step, long long capacity, double* __restrict__ last) {
  __shared__ double red[kBlock / 64];
  __shared__ bool s_last;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  double ke = 0.0;
  if (i < n) {
    const double h = hdt_m[i], m = mass[i];
    for (int d = 0; d < 3; d++) {
      const double pv = fma(h, f[3 * i + d], v[3 * i + d]);
      v[3 * i + d] = pv;
      ke = fma(0.5 * m * pv, pv, ke);
    }
  }
  ke = block_sum(ke, red);
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&acc[0], ke, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  // tether energy: the pre kernel's per-block partials in fixed order (its grid is this kernel's)
  double et = 0.0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += kBlock) et += tether_part[b];
  et = block_sum(et, red);
  if (threadIdx.x == 0) {
    __threadfence();
    const double kin = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double pot = et + energy[0];  // (what the AGBNP evaluation added to the word since it was last handed back as zero)
    const long long s = step[0];
    if (s < capacity) log_pe[s] = pot, log_ke[s] = kin;
    last[0] = pot, last[1] = kin;
    step[0] = s + 1;
    energy[0] = 0.0;
    acc[0] = 0.0;
    *done = 0u;
  }
}

// Between two force evaluations of a run of steps: the back half of step n (second kick, energies logged) and the front
// half of step n + 1 in ONE launch.  The tether partials are double-buffered (the workgroup that arrives last sums step n's
// while the others already write step n + 1's): part_old is read, part_new written.
__global__ __launch_bounds__(kBlock) void k_md_mid(int n, int kind, double* __restrict__ x, double* __restrict__ v, double* __restrict__ f,
                                                  const double* __restrict__ x0, const double* __restrict__ hdt_m, const double* __restrict__ mass,
                                                  const double* __restrict__ c2, double c1, double dt, double ktether, unsigned long long seed,
                                                  double* __restrict__ energy, const double* __restrict__ part_old, double* __restrict__ part_new,
                                                  double* __restrict__ acc, unsigned* __restrict__ done, double* __restrict__ log_pe,
                                                  double* __restrict__ log_ke, long long* __restrict__ step, long long capacity,
                                                  double* __restrict__ last) {
  __shared__ double red[kBlock / 64];
  __shared__ bool s_last;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long s = (unsigned long long)step[0];  // (step n: read before this workgroup counts itself in)
  double ke = 0.0, e = 0.0;
  if (i < n) {
    const double h = hdt_m[i], m = mass[i];
    double px[3], pv[3];
    for (int d = 0; d < 3; d++) {
      const double fd = f[3 * i + d];
      const double v1 = fma(h, fd, v[3 * i + d]);  // end of step n
      ke = fma(0.5 * m * v1, v1, ke);
      pv[d] = fma(h, fd, v1);                       // first kick of step n + 1: the same force
      px[d] = x[3 * i + d];
    }
    e = front_half(i, kind, px, pv, x, v, f, x0, c2, c1, dt, ktether, seed, s + 1);
  }
  ke = block_sum(ke, red);
  e = block_sum(e, red);
  if (threadIdx.x == 0) {
    part_new[blockIdx.x] = e;
    __hip_atomic_fetch_add(&acc[0], ke, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  double et = 0.0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += kBlock) et += part_old[b];
  et = block_sum(et, red);
  if (threadIdx.x == 0) {
    __threadfence();
    const double kin = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double pot = et + energy[0];
    if ((long long)s < capacity) log_pe[s] = pot, log_ke[s] = kin;
    last[0] = pot, last[1] = kin;
    step[0] = (long long)s + 1;
    energy[0] = 0.0;
    acc[0] = 0.0;
    *done = 0u;
  }
}

// tethers alone (the first force evaluation of a run, and the minimiser's): f = -k (x - x0), partials of their energy
__global__ __launch_bounds__(kBlock) void k_md_tethers(int n, const double* __restrict__ x, const double* __restrict__ x0, double* __restrict__ f,
                                                      double ktether, double* __restrict__ tether_part) {
  __shared__ double red[kBlock / 64];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  double e = 0.0;
  if (i < n)
    for (int d = 0; d < 3; d++) {
      const double dd = x[3 * i + d] - x0[3 * i + d];
      f[3 * i + d] = -ktether * dd;
      e = fma(0.5 * ktether * dd, dd, e);
    }
  e = block_sum(e, red);
  if (threadIdx.x == 0) tether_part[blockIdx.x] = e;
}

}  // namespace

extern "C" {

int agbnp_md_blocks(int n) { return (n + kBlock - 1) / kBlock; }

int agbnp_md_pre(int n, int kind, double* x, double* v, double* f, const double* x0, const double* hdt_m, const double* c2, double c1,
                 double dt, double ktether, unsigned long long seed, const long long* step, double* tether_part, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_md_pre, dim3(agbnp_md_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, n, kind, x, v, f, x0, hdt_m, c2, c1, dt, ktether,
                     seed, step, tether_part);
  return (int)hipGetLastError();
}

int agbnp_md_post(int n, double* v, const double* f, const double* hdt_m, const double* mass, double* energy, const double* tether_part,
                  double* acc, unsigned* done, double* log_pe, double* log_ke, long long* step, long long capacity, double* last,
                  void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_md_post, dim3(agbnp_md_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, n, v, f, hdt_m, mass, energy, tether_part, acc,
                     done, log_pe, log_ke, step, capacity, last);
  return (int)hipGetLastError();
}

int agbnp_md_mid(int n, int kind, double* x, double* v, double* f, const double* x0, const double* hdt_m, const double* mass, const double* c2,
                 double c1, double dt, double ktether, unsigned long long seed, double* energy, const double* part_old, double* part_new,
                 double* acc, unsigned* done, double* log_pe, double* log_ke, long long* step, long long capacity, double* last, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_md_mid, dim3(agbnp_md_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, n, kind, x, v, f, x0, hdt_m, mass, c2, c1, dt, ktether,
                     seed, energy, part_old, part_new, acc, done, log_pe, log_ke, step, capacity, last);
  return (int)hipGetLastError();
}

int agbnp_md_tethers(int n, const double* x, const double* x0, double* f, double ktether, double* tether_part, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_md_tethers, dim3(agbnp_md_blocks(n)), dim3(kBlock), 0, (hipStream_t)stream, n, x, x0, f, ktether, tether_part);
  return (int)hipGetLastError();
}

}  // extern "C"
