// Pair stages of AGBNP1 on gfx950: inverse Born radii, GB pair energy/forces, Born-radius chain-rule
// forces, plus the per-atom glue between them.
//
// Per-atom glue (Born radius from the descreening sum, bru from Y, nu from W+U) has no kernels of its own: the
// consumer recomputes it from the finished sums in its prologue, which costs a few dozen flops per atom and
// saves three launches per evaluation.
//
// Reference semantics (platforms/reference/src/ReferenceAGBNPKernels.cpp, restated in oracle run_v1()):
//   :420-433  volume scaling factors s_i = selfvol_i / (4 pi R_i^3 / 3)
//   :435-454  beta_i = 1/R_i - (1/4pi) sum_{j heavy, j!=i, d<2nm} s_j Q(d; type_i, type_j);  B_i, f'_i (:41-55)
//   :464-504  GB pair energy, direct pair force, Y_i                         (ALL pairs, no cutoff)
//   :513-542  van der Waals energy, brw_i, bru_i
//   :555-586  chain-rule forces through the Born radii and the U_j / W_j sums
//
// Machine mapping: "row form".  A wavefront owns 64 consecutive i-atoms (one per lane, everything in
// registers) and walks a contiguous range of j-atoms whose records are wave-uniform, so they arrive
// through the scalar cache (s_load) and cost no vector memory traffic or LDS; the j range is split
// over blockIdx.y to fill the chip, each split writes its own partial row and the next per-atom kernel
// adds the partial rows in a fixed order (no atomics -> bit-reproducible).  Every pair is evaluated
// from both ends instead of scattering the reaction force; the I4 spline tables sit in LDS.
#include <hip/hip_runtime.h>

#include "agbnp_common.h"
#include "pair_kernels.h"

namespace agbnp {

constexpr int kPairWaves = 4;               // waves per pair-kernel workgroup; all work on the same 64 i-atoms
constexpr int kPairBlock = 64 * kPairWaves;

// ---- I4 spline (uniform nodes x_k = k*dr, k = 0..15; table entry = {y_k, y2_k*dr^2/6}) ------------------
__device__ __forceinline__ double spline_value(const double2* __restrict__ tab, int base, double d) {
  const double t = d * ((kI4Nodes - 1) / kI4MaxA);
  int k = (int)t;
  k = k > kI4Nodes - 2 ? kI4Nodes - 2 : k;
  const double a = (double)(k + 1) - t;
  const double b = 1.0 - a;
  const double2 lo = tab[base + k], hi = tab[base + k + 1];
  return a * lo.x + b * hi.x + (a * a * a - a) * lo.y + (b * b * b - b) * hi.y;
}
__device__ __forceinline__ void spline_value_deriv(const double2* __restrict__ tab, int base, double d, double& val, double& der) {
  const double invdr = (kI4Nodes - 1) / kI4MaxA;
  const double t = d * invdr;
  int k = (int)t;
  k = k > kI4Nodes - 2 ? kI4Nodes - 2 : k;
  const double a = (double)(k + 1) - t;
  const double b = 1.0 - a;
  const double2 lo = tab[base + k], hi = tab[base + k + 1];
  val = a * lo.x + b * hi.x + (a * a * a - a) * lo.y + (b * b * b - b) * hi.y;
  der = ((hi.x - lo.x) + (1.0 - 3.0 * a * a) * lo.y + (3.0 * b * b - 1.0) * hi.y) * invdr;
}

__device__ __forceinline__ void hbm_add(double* p, double v) {  // global_atomic_add_f64
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- geometry in, accumulators cleared -------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(PairArgs P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kStatWords && blockIdx.x == 0) P.status[i] = 0;
  {
    // bounding box of every block of 64 consecutive atoms (one wave each) for the tile culling of the
    // range-limited pair stages
    const int ic = i < P.n ? i : P.n - 1;
    double lo[3], hi[3];
    for (int d = 0; d < 3; d++) lo[d] = hi[d] = P.pos[3 * ic + d];
    for (int off = 32; off > 0; off >>= 1)
      for (int d = 0; d < 3; d++) {
        lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64));
        hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64));
      }
    const int blk = i >> 6;
    if ((i & 63) == 0 && blk * 64 < P.n) {
      for (int d = 0; d < 3; d++) {
        P.abox[6 * blk + d] = lo[d];
        P.abox[6 * blk + 3 + d] = hi[d];
      }
    }
  }
  if (i >= P.n) return;
  const double x = P.pos[3 * i], y = P.pos[3 * i + 1], z = P.pos[3 * i + 2];
  P.aposq[i] = make_double4(x, y, z, P.charge[i]);
  P.gb_fx[i] = 0.0;  // GB sums arrive through atomics
  P.gb_fy[i] = 0.0;
  P.gb_fz[i] = 0.0;
  P.gb_y[i] = 0.0;
  P.born_part[i] = 0.0;  // Born sums and chain-rule sums arrive through atomics too
  P.db_fx[i] = 0.0;
  P.db_fy[i] = 0.0;
  P.db_fz[i] = 0.0;
  P.db_wu[i] = 0.0;
  const int h = P.a2h[i];
  if (h >= 0) {
    P.hx[h] = x;
    P.hy[h] = y;
    P.hz[h] = z;
    P.gx[h] = 0.0;
    P.gy[h] = 0.0;
    P.gz[h] = 0.0;
    P.sv_vdw[h] = 0.0;
    P.sv_large[h] = 0.0;
    P.gam[h] = P.gam_cav[h];
  }
}

// ---- volume scaling factors (ReferenceAGBNPKernels.cpp:420-433) packed with the heavy positions ---------
__global__ __launch_bounds__(256) void k_scale(PairArgs P) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= P.nh) return;
  const double sj = P.sv_vdw[h] * P.inv_vol_h[h];
  P.hposs[h] = make_double4(P.hx[h], P.hy[h], P.hz[h], sj);
  P.scale[P.h2a[h]] = sj;  // hydrogens keep the 0 they were created with
}

// ---- inverse Born radii: partial sums over a j range ---------------------------------------------------
__global__ __launch_bounds__(kPairBlock) void k_born_pairs(int n, int nh, int hchunk, int ntj, int lut_entries,
                                                   const double4* __restrict__ aposq, const int2* __restrict__ ameta,
                                                   const double4* __restrict__ hposs, const int2* __restrict__ hmeta,
                                                   const double2* __restrict__ lut, double* __restrict__ born_part) {
  extern __shared__ double2 s_lut[];
  __shared__ double s_red[kPairWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < lut_entries; t += kPairBlock) s_lut[t] = lut[t];
  __syncthreads();
  const int i = blockIdx.x * 64 + lane;
  const bool valid = i < n;
  const int ii = valid ? i : n - 1;
  const double4 pi = aposq[ii];
  const int tbase = ameta[ii].x * ntj;
  // the block's j range is split once more over its waves (same 64 i-atoms in every wave)
  const int sub = (hchunk + kPairWaves - 1) / kPairWaves;
  const int jb = blockIdx.y * hchunk;
  const int j0 = min(nh, jb + wave * sub);
  const int j1 = min(min(nh, jb + hchunk), j0 + sub);
  double sum = 0.0;
  auto pair = [&](const double4& pj, const int2& mj) {
    const double dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (d2 < kI4MaxA * kI4MaxA && mj.x != i) {
      const double d = sqrt(d2);
      sum += pj.w * spline_value(s_lut, (tbase + mj.y) * kI4Nodes, d);
    }
  };
  if (j0 < j1) {
    // Scalar-load pipeline.  SMEM returns out of order, so only lgkmcnt(0) is a safe wait: retire the record of
    // j (in flight since the previous half-iteration) BEFORE issuing the loads of j+1, then compute j while j+1
    // is in flight.  The empty asm consumes the registers (forces the wait there), sched_barrier pins the order.
    double4 pA = hposs[j0];
    int2 mA = hmeta[j0];
    int j = j0;
    for (; j + 1 < j1; j += 2) {
      asm volatile("; j landed" ::"s"(pA.x), "s"(mA.x));
      const double4 pB = hposs[j + 1];
      const int2 mB = hmeta[j + 1];
      __builtin_amdgcn_sched_barrier(0);
      pair(pA, mA);
      asm volatile("; j+1 landed" ::"s"(pB.x), "s"(mB.x));
      const int jn = j + 2 < j1 ? j + 2 : j + 1;
      pA = hposs[jn];
      mA = hmeta[jn];
      __builtin_amdgcn_sched_barrier(0);
      pair(pB, mB);
    }
    if (j < j1) pair(pA, mA);
  }
  s_red[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && valid) {
    double t = s_red[0][lane];
    for (int w = 1; w < kPairWaves; w++) t += s_red[w][lane];
    hbm_add(&born_part[i], t);  // single row: the per-atom kernel that follows reads one value
  }
}

// ---- per-atom Born-radius algebra, recomputed by its consumers instead of a kernel of its own ---------------
// beta_i = 1/R_i - (1/4pi) sum_j s_j Q  ->  B_i = 1/f(beta_i), f' (ReferenceAGBNPKernels.cpp:41-55,450-454)
struct BornRadius {
  double br, inv_br, fp;
};
__device__ __forceinline__ BornRadius born_radius(double inv_rvdw, double qsum) {
  const double beta = inv_rvdw - (1. / (4. * kPi)) * qsum;
  const double amin = 1. / kI4MaxA, a2 = 1. / (kI4MaxA * kI4MaxA);
  BornRadius r;
  if (beta < 0.0) {
    r.inv_br = amin;
    r.fp = 0.0;
  } else {
    r.inv_br = sqrt(a2 + beta * beta);
    r.fp = beta / r.inv_br;
  }
  r.br = 1. / r.inv_br;
  return r;
}

// ---- GB pairs, symmetric 64x64 tiles (all pairs, no cutoff) ------------------------------------------------
// A wave owns one half of a tile (I <= J): lane l keeps atom i = 64 I + l and its sums in registers, a second
// register set (record + sums of one j atom of block J) travels round the wave by DPP wave rotation, so every
// (i, j) pair of the tile meets exactly once and both ends are updated from one evaluation of the pair terms
// (half the FP64 work of the row form; no LDS, no vector memory in the loop).  Off-diagonal tiles are cut in
// four work items of 16 rotations; a diagonal tile is two items that visit cyclic distances 1..32 (distance 32
// only from the lower half of the lanes).  Per-atom sums leave through FP64 HBM atomics into single rows.
constexpr int kGbSteps = AGBNP_GB_STEPS;  // rotations per work item: 64 = a whole off-diagonal tile, 32 = half, 16 = quarter

__device__ __forceinline__ double rot1(double v) {  // lane l <- lane l+1 (mod 64)
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x134, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x134, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(64) void k_gb_tiles(int n, const int* __restrict__ items, const double4* __restrict__ aposq,
                                                 const double* __restrict__ born_part, const double* __restrict__ inv_rvdw,
                                                 const double* __restrict__ alpha, double* __restrict__ born,
                                                 double* __restrict__ born_fp, double* __restrict__ brw,
                                                 double* __restrict__ e_atom, double* __restrict__ gb_fx,
                                                 double* __restrict__ gb_fy, double* __restrict__ gb_fz,
                                                 double* __restrict__ gb_y, double* __restrict__ egb_part) {
  const int lane = threadIdx.x;
  const int item = items[blockIdx.x];
  const int I = item & 0xfff, J = (item >> 12) & 0xfff, part = (item >> 24) & 3;
  const bool diag = I == J;
  const int nsteps = diag ? (kGbSteps < 32 ? kGbSteps : 32) : kGbSteps;
  const int start = (diag ? 1 : 0) + nsteps * part;  // cyclic offset of the first j met by lane l
  const int i = 64 * I + lane;
  const bool vi = i < n;
  const int ic = vi ? i : n - 1;
  const double4 pi = aposq[ic];
  // Born radii from the finished descreening sums (every item recomputes them for its 128 atoms: a few dozen
  // flops per atom against 32 x 64 pair evaluations, and one kernel launch less per evaluation)
  const BornRadius bri = born_radius(inv_rvdw[ic], born_part[ic]);
  const double2 bi = make_double2(bri.br, bri.inv_br);
  const double qi = vi ? pi.w : 0.0;
  if (diag && part == 0 && vi) {
    // the diagonal item of a block publishes the per-atom results exactly once:
    // B_i, f'_i, vdW energy + GB self energy, brw_i (ReferenceAGBNPKernels.cpp:477,513-533)
    const double bh = bri.br + kHBRadius, bh3 = bh * bh * bh, al = alpha[i];
    born[i] = bri.br;
    born_fp[i] = bri.fp;
    e_atom[i] = al / bh3 + kDielFactor * pi.w * pi.w * bri.inv_br;
    brw[i] = -(1. / (4. * kPi)) * 3. * al * bri.br * bri.br * bri.fp / (bh3 * bh);
  }
  const int j = 64 * J + ((lane + start) & 63);
  const bool vj = j < n;
  const int jc = vj ? j : n - 1;
  const double4 pj0 = aposq[jc];
  const BornRadius brj = born_radius(inv_rvdw[jc], born_part[jc]);
  double xj = pj0.x, yj = pj0.y, zj = pj0.z, qj = vj ? pj0.w : 0.0, bj = brj.br, ibj = brj.inv_br;
  double fxi = 0, fyi = 0, fzi = 0, yi = 0, fxj = 0, fyj = 0, fzj = 0, yj_acc = 0, e = 0;
#pragma unroll 2
  for (int k = 0; k < nsteps; k++) {
    const double dx = xj - pi.x, dy = yj - pi.y, dz = zj - pi.z;
    const double d2 = dx * dx + dy * dy + dz * dz;
    const double bb = bi.x * bj;
    const double et = exp(-0.25 * d2 * (bi.y * ibj));  // exp(-d^2 / (4 B_i B_j))
    const double fgb = rsqrt(d2 + bb * et);
    const double fgb3 = fgb * fgb * fgb;
    // diagonal tile, cyclic distance 32: the pair (l, l+32) would otherwise be met from both ends
    const double qqf = (diag && start + k == 32 && lane >= 32) ? 0.0 : qi * qj;
    const double qq = kDielFactor * qqf;
    e += 2.0 * qq * fgb;
    const double mw = -2.0 * qq * (1.0 - 0.25 * et) * fgb3;
    const double gx = dx * mw, gy = dy * mw, gz = dz * mw;
    fxi += gx;
    fyi += gy;
    fzi += gz;
    fxj -= gx;
    fyj -= gy;
    fzj -= gz;
    const double yt = qqf * (bb + 0.25 * d2) * et * fgb3;
    yi += yt;
    yj_acc += yt;
    xj = rot1(xj);
    yj = rot1(yj);
    zj = rot1(zj);
    qj = rot1(qj);
    bj = rot1(bj);
    ibj = rot1(ibj);
    fxj = rot1(fxj);
    fyj = rot1(fyj);
    fzj = rot1(fzj);
    yj_acc = rot1(yj_acc);
  }
  if (vi) {
    hbm_add(&gb_fx[i], fxi);
    hbm_add(&gb_fy[i], fyi);
    hbm_add(&gb_fz[i], fzi);
    hbm_add(&gb_y[i], yi);
  }
  const int jend = 64 * J + ((lane + start + nsteps) & 63);  // whose sums this lane holds after the rotations
  if (jend < n) {
    hbm_add(&gb_fx[jend], fxj);
    hbm_add(&gb_fy[jend], fyj);
    hbm_add(&gb_fz[jend], fzj);
    hbm_add(&gb_y[jend], yj_acc);
  }
  e = wave_sum(e);
  if (lane == 0) egb_part[blockIdx.x] = e;
}

// ---- Born-radius chain rule, symmetric 64x64 tiles with range culling -------------------------------------
// Reference loop (ReferenceAGBNPKernels.cpp:555-586) over ordered (i, heavy j != i, d < 2 nm):
//   W_j += brw_i Q,  U_j += bru_i Q,  F_i += D (brw_i + bru_i) s_j Q'/d,  F_j -= same      (D = r_j - r_i)
// Same machinery as k_gb_tiles: block I in registers, the record and sums of block J travel by DPP rotation,
// every unordered pair is met once and serves both directions (two table look-ups, one distance).
// A work item whose two 64-atom bounding boxes are more than the table's 2 nm reach apart exits at once.
__device__ __forceinline__ int rot1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false); }

__global__ __launch_bounds__(256) void k_dborn_tiles(int n, int ntj, int lut_entries, int nitems, const int* __restrict__ items,
                                                    const double* __restrict__ abox, const double4* __restrict__ aposq,
                                                    const int2* __restrict__ ameta, const double* __restrict__ born,
                                                    const double* __restrict__ born_fp, const double* __restrict__ brw,
                                                    const double* __restrict__ gb_y, const double* __restrict__ scale,
                                                    const double2* __restrict__ lut, double* __restrict__ db_fx,
                                                    double* __restrict__ db_fy, double* __restrict__ db_fz,
                                                    double* __restrict__ db_wu) {
  extern __shared__ double2 s_lut[];
  // four waves = four work items share one copy of the spline tables
  const int lane = threadIdx.x & 63;
  const int item_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int item = items[item_id < nitems ? item_id : nitems - 1];
  const int I = item & 0xfff, J = (item >> 12) & 0xfff, part = (item >> 24) & 3;
  const bool diag = I == J;
  bool in_range = item_id < nitems;
  if (!diag) {  // wave-uniform range test on the two bounding boxes
    double gap2 = 0.0;
    for (int d = 0; d < 3; d++) {
      const double g = fmax(0.0, fmax(abox[6 * J + d] - abox[6 * I + 3 + d], abox[6 * I + d] - abox[6 * J + 3 + d]));
      gap2 += g * g;
    }
    in_range = in_range && gap2 < kI4MaxA * kI4MaxA;
  }
  if (__syncthreads_or(in_range ? 1 : 0) == 0) return;  // the whole workgroup is out of range
  for (int t = threadIdx.x; t < lut_entries; t += 256) s_lut[t] = lut[t];
  __syncthreads();
  if (!in_range) return;
  const int nsteps = diag ? (kGbSteps < 32 ? kGbSteps : 32) : kGbSteps;
  const int start = (diag ? 1 : 0) + nsteps * part;
  const int i = 64 * I + lane;
  const bool vi = i < n;
  // {bw, s} of an atom: bw = brw + bru with bru = -(1/4pi) k (q^2 + Y B) f' (ReferenceAGBNPKernels.cpp:534-542),
  // formed here from the finished GB sums instead of a per-atom kernel in between
  auto weights = [&](int a, double q) {
    const double bru = -(1. / (4. * kPi)) * kDielFactor * (q * q + gb_y[a] * born[a]) * born_fp[a];
    return make_double2(brw[a] + bru, scale[a]);
  };
  const double4 pi = aposq[vi ? i : n - 1];
  const double2 wi = vi ? weights(i, pi.w) : make_double2(0.0, 0.0);  // zero weights switch a padded lane off
  const int2 mi = ameta[vi ? i : n - 1];                    // {screened type, screener type or -1}
  const int tsr_i = vi ? mi.y : -1;
  const int j = 64 * J + ((lane + start) & 63);
  const bool vj = j < n;
  const double4 pj0 = aposq[vj ? j : n - 1];
  const double2 wj0 = vj ? weights(j, pj0.w) : make_double2(0.0, 0.0);
  const int2 mj0 = ameta[vj ? j : n - 1];
  double xj = pj0.x, yj = pj0.y, zj = pj0.z, bwj = wj0.x, sj = wj0.y;
  int tj = mj0.x | (((vj ? mj0.y : -1) + 1) << 16);  // screened type | (screener type + 1) << 16
  int jid = vj ? j : -1;
  double fxi = 0, fyi = 0, fzi = 0, wui = 0, fxj = 0, fyj = 0, fzj = 0, wuj = 0;
#pragma unroll 2
  for (int k = 0; k < nsteps; k++) {
    const double dx = xj - pi.x, dy = yj - pi.y, dz = zj - pi.z;
    const double d2 = dx * dx + dy * dy + dz * dz;
    const bool once = !(diag && start + k == 32 && lane >= 32);  // diagonal tile, distance 32: one end only
    if (d2 < kI4MaxA * kI4MaxA && vi && jid >= 0 && once) {
      const double rinv = rsqrt(d2);
      const double d = d2 * rinv;
      const int tsd_j = tj & 0xffff, tsr_j = (tj >> 16) - 1;
      double t = 0.0;
      if (tsr_j >= 0) {  // j descreens i
        double q1, dq1;
        spline_value_deriv(s_lut, (mi.x * ntj + tsr_j) * kI4Nodes, d, q1, dq1);
        wuj += wi.x * q1;
        t += wi.x * sj * dq1;
      }
      if (tsr_i >= 0) {  // i descreens j
        double q2, dq2;
        spline_value_deriv(s_lut, (tsd_j * ntj + tsr_i) * kI4Nodes, d, q2, dq2);
        wui += bwj * q2;
        t += bwj * wi.y * dq2;
      }
      t *= rinv;
      const double gx = dx * t, gy = dy * t, gz = dz * t;
      fxi += gx;
      fyi += gy;
      fzi += gz;
      fxj -= gx;
      fyj -= gy;
      fzj -= gz;
    }
    xj = rot1(xj);
    yj = rot1(yj);
    zj = rot1(zj);
    bwj = rot1(bwj);
    sj = rot1(sj);
    tj = rot1i(tj);
    jid = rot1i(jid);
    fxj = rot1(fxj);
    fyj = rot1(fyj);
    fzj = rot1(fzj);
    wuj = rot1(wuj);
  }
  if (vi) {
    hbm_add(&db_fx[i], fxi);
    hbm_add(&db_fy[i], fyi);
    hbm_add(&db_fz[i], fzi);
    hbm_add(&db_wu[i], wui);
  }
  if (jid >= 0) {  // after the rotations the lane holds the sums of atom jid
    hbm_add(&db_fx[jid], fxj);
    hbm_add(&db_fy[jid], fyj);
    hbm_add(&db_fz[jid], fzj);
    hbm_add(&db_wu[jid], wuj);
  }
}

// ---- outputs: one launch, three concurrent roles ---------------------------------------------------------
//   blocks [0, nfb)  forces: F = -grad(tree) + sum of the pair partial rows, ADDED to the caller's buffer
//   block  nfb       energy: fixed-order sum of every energy partial, ADDED to the caller's scalar
//   block  nfb+1     bookkeeping for the NEXT evaluation: tree statistics and the largest-first subtree order
__device__ __forceinline__ double block_sum_256(double v, double* red4) {
  v = wave_sum(v);
  const int t = threadIdx.x;
  if ((t & 63) == 0) red4[t >> 6] = v;
  __syncthreads();
  const double r = (red4[0] + red4[1]) + (red4[2] + red4[3]);  // fixed order -> reproducible
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void k_outputs(PairArgs P, int version, double* __restrict__ force_out,
                                                 double* __restrict__ energy_out, double* __restrict__ components) {
  const int nfb = (P.n + 255) / 256;
  const int t = threadIdx.x;
  if ((int)blockIdx.x < nfb) {
    const int i = blockIdx.x * 256 + t;
    if (i >= P.n) return;
    double fx = 0, fy = 0, fz = 0;
    const int h = P.a2h[i];
    if (h >= 0) {  // cavity + pseudo-volume gradients -> force
      fx = -P.gx[h];
      fy = -P.gy[h];
      fz = -P.gz[h];
    }
    if (version == 1) {
      fx += P.gb_fx[i] + P.db_fx[i];
      fy += P.gb_fy[i] + P.db_fy[i];
      fz += P.gb_fz[i] + P.db_fz[i];
    }
    force_out[3 * i] += fx;
    force_out[3 * i + 1] += fy;
    force_out[3 * i + 2] += fz;
    return;
  }
  if ((int)blockIdx.x == nfb) {
    __shared__ double red4[4];
    // strided partial sums with 8 independent loads in flight per thread (a dependent load per trip would
    // cost one HBM/L2 latency each); the per-thread order is fixed, so the result is reproducible
    auto strided_sum = [&](const double* __restrict__ a, int count, int stride, int offset) {
      double acc = 0.0;
      for (int base = 0; base < count; base += 256 * 8) {
        double v[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const int k = base + b * 256 + t;
          v[b] = k < count ? a[(size_t)k * stride + offset] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < 8; b++) acc += v[b];
      }
      return acc;
    };
    const double ecav1 = strided_sum(P.epart, P.nh, 2, 0), ecav2 = strided_sum(P.epart, P.nh, 2, 1);
    double eatom = 0, egb = 0;
    if (version == 1) {
      eatom = strided_sum(P.e_atom, P.n, 1, 0);
      egb = strided_sum(P.egb_part, P.egb_parts, 1, 0);
    }
    const double o0 = block_sum_256(ecav1, red4), o1 = block_sum_256(ecav2, red4);
    const double o2 = block_sum_256(eatom, red4), o3 = block_sum_256(egb, red4);
    if (t == 0) {
      components[0] = o0;
      components[1] = o1;
      components[2] = o2;
      components[3] = o3;
      energy_out[0] += o0 + o1 + o2 + o3;
    }
    return;
  }
  // ---- bookkeeping block
  constexpr int kBins = 512, kBatch = 8;
  __shared__ int hist[kBins], start[kBins], part[4], imax[8];
  for (int k = t; k < kBins; k += 256) hist[k] = 0;
  __syncthreads();
  int tot = 0, mx = 0, ma = 0;
  for (int base = 0; base < P.nh; base += 256 * kBatch) {
    int2 sz[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; b++) {  // independent loads first, then the (slow) LDS atomics
      const int h = base + b * 256 + t;
      sz[b] = h < P.nh ? P.sizes[h] : make_int2(-1, 0);
    }
#pragma unroll
    for (int b = 0; b < kBatch; b++) {
      if (sz[b].x >= 0) {
        tot += sz[b].x;
        mx = sz[b].x > mx ? sz[b].x : mx;
        ma = sz[b].y > ma ? sz[b].y : ma;
        const int key = kBins - 1 - (sz[b].x >> 2);
        atomicAdd(&hist[key < 0 ? 0 : key], 1);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    tot += __shfl_xor(tot, off, 64);
    mx = max(mx, __shfl_xor(mx, off, 64));
    ma = max(ma, __shfl_xor(ma, off, 64));
  }
  if ((t & 63) == 0) {
    part[t >> 6] = tot;
    imax[t >> 6] = mx;
    imax[4 + (t >> 6)] = ma;
  }
  __syncthreads();
  if (t == 0) {
    P.status[kStatTotalNodes] = part[0] + part[1] + part[2] + part[3];
    P.status[kStatMaxNodes] = max(max(imax[0], imax[1]), max(imax[2], imax[3]));
    P.status[kStatMaxAtoms] = max(max(imax[4], imax[5]), max(imax[6], imax[7]));
  }
  // exclusive scan of the histogram (bins are in descending size order): thread t owns bins 2t, 2t+1
  const int h0 = hist[2 * t], h1 = hist[2 * t + 1];
  int incl = h0 + h1;
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if ((t & 63) >= off) incl += v;
  }
  __syncthreads();
  if ((t & 63) == 63) part[t >> 6] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < (t >> 6); w++) before += part[w];
  const int excl = before + incl - (h0 + h1);
  start[2 * t] = excl;
  start[2 * t + 1] = excl + h0;
  __syncthreads();
  // largest-first processing order of the next evaluation (geometry changes little between MD steps, so this
  // step's sizes predict the next step's work)
  for (int base = 0; base < P.nh; base += 256 * kBatch) {
    int nn[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; b++) {
      const int h = base + b * 256 + t;
      nn[b] = h < P.nh ? P.sizes[h].x : -1;
    }
#pragma unroll
    for (int b = 0; b < kBatch; b++) {
      if (nn[b] >= 0) {
        const int key = kBins - 1 - (nn[b] >> 2);
        const int pos = atomicAdd(&start[key < 0 ? 0 : key], 1);
        P.order[pos] = base + b * 256 + t;
      }
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------------
#define AGBNP_CHECK_LAUNCH()             \
  do {                                   \
    hipError_t e__ = hipGetLastError();  \
    if (e__ != hipSuccess) return e__;   \
  } while (0)

#define AGBNP_MARK(id)                               \
  do {                                               \
    if (tl) {                                        \
      hipError_t m__ = tl->mark(id, st);             \
      if (m__ != hipSuccess) return m__;             \
    }                                                \
  } while (0)

hipError_t launch_prep(const PairArgs& P, hipStream_t st, Timeline* tl) {
  AGBNP_MARK(kKPrep);
  const int n = P.n > kStatWords ? P.n : kStatWords;
  hipLaunchKernelGGL(k_prep, dim3((n + 255) / 256), dim3(256), 0, st, P);
  return hipGetLastError();
}

hipError_t launch_pair_stages(const PairArgs& P, hipStream_t st, Timeline* tl) {
  const int nblk = (P.n + 63) / 64;
  const size_t lds = (size_t)P.lut_entries * sizeof(double2);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_born_pairs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dborn_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  AGBNP_MARK(kKScale);
  hipLaunchKernelGGL(k_scale, dim3((P.nh + 255) / 256 > 0 ? (P.nh + 255) / 256 : 1), dim3(256), 0, st, P);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(kKBornPairs);
  hipLaunchKernelGGL(k_born_pairs, dim3(nblk, P.hsplits), dim3(kPairBlock), lds, st, P.n, P.nh, P.hchunk, P.ntj, P.lut_entries,
                     (const double4*)P.aposq, P.ameta, (const double4*)P.hposs, P.hmeta, P.lut, P.born_part);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(kKGbTiles);
  hipLaunchKernelGGL(k_gb_tiles, dim3(P.gb_items_count), dim3(64), 0, st, P.n, P.gb_items, (const double4*)P.aposq,
                     (const double*)P.born_part, P.inv_rvdw, P.alpha, P.born, P.born_fp, P.brw, P.e_atom, P.gb_fx, P.gb_fy,
                     P.gb_fz, P.gb_y, P.egb_part);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(kKDbornTiles);
  hipLaunchKernelGGL(k_dborn_tiles, dim3((P.gb_items_count + 3) / 4), dim3(256), lds, st, P.n, P.ntj, P.lut_entries,
                     P.gb_items_count, P.gb_items, (const double*)P.abox, (const double4*)P.aposq, P.ameta,
                     (const double*)P.born, (const double*)P.born_fp, (const double*)P.brw, (const double*)P.gb_y,
                     (const double*)P.scale, P.lut, P.db_fx, P.db_fy, P.db_fz, P.db_wu);
  AGBNP_CHECK_LAUNCH();
  return hipSuccess;
}

hipError_t launch_outputs(const PairArgs& P, int version, double* force_out, double* energy_out, double* components, hipStream_t st,
                          Timeline* tl) {
  AGBNP_MARK(kKOutputs);
  hipLaunchKernelGGL(k_outputs, dim3((P.n + 255) / 256 + 2), dim3(256), 0, st, P, version, force_out, energy_out, components);
  AGBNP_CHECK_LAUNCH();
  AGBNP_MARK(-1);
  return hipSuccess;
}

}  // namespace agbnp
