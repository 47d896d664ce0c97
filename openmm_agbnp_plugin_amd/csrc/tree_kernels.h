// GaussVol overlap-tree kernels for gfx950 (MI355X).
//
// One 64-lane wavefront owns the complete overlap subtree rooted at one heavy atom and keeps it in
// LDS for its whole life: build (large radii) -> bottom-up volume pass -> top-down rescan with vdW
// radii -> second bottom-up pass, all inside one launch.  Only per-atom sums (gradients, self
// volumes), one energy pair per subtree and the 8-byte/node topology leave the CU.
//
// What is computed (reference restated in oracle/agbnp_oracle.cpp):
//   build      : gaussvol/gaussvol.cpp:197-250 (child scan over YOUNGER siblings), :154-192 (children sorted
//                by switched volume, descending), :376-397 (tree growth), MAX_ORDER=8 stop (:211)
//   merge      : gaussvol/gaussvol.cpp:60-93 (ogauss_alpha), :18-41 (pol_switchfunc)
//   bottom-up  : gaussvol/gaussvol.cpp:400-487 (compute_volume_underslot2_r), only the quantities the
//                Reference kernel consumes: self volumes, energy, energy gradient
//   rescan     : gaussvol/gaussvol.cpp:254-327 (volumes) and :330-372 (gammas only)
//
// Design differences from the reference (same numbers, different machine):
//   * breadth-first node order inside a subtree (levels contiguous) so that both sweeps are
//     lane-parallel over a level; the reference is depth-first recursive.
//   * per-node derived quantities (switched volume, sfp, dv1, dvv1) are recomputed from
//     (parent Gaussian, atom Gaussian) in the sweep instead of being stored: 7 doubles/node in LDS.
//   * child compaction = wave ballot + mbcnt prefix; child ordering = in-register rank sort with
//     v_readlane broadcasts (no LDS traffic, no barrier) for up to 64 candidates.
#pragma once
#include <hip/hip_runtime.h>

#include "agbnp_common.h"

namespace agbnp {

struct TreeArgs {
  int nh;  // heavy atoms
  const double *hx, *hy, *hz;  // heavy-atom positions (SoA, heavy index)
  const double *a_large, *v_large, *a_vdw, *v_vdw;  // Gaussian exponent / volume per heavy atom
  const double* gam;  // per heavy atom: gamma/roffset (pass 1 uses +gam, pass 2 uses -gam); pass 3: (W+U)/V_vdw
  double rcut2;       // conservative squared cutoff of the 2-body overlap search
  double* gx;         // [nh] gradient accumulators (dE/dr), heavy index
  double* gy;
  double* gz;
  double* sv_large;   // [nh] self volumes with enlarged radii (diagnostic; may be null)
  double* sv_vdw;     // [nh] self volumes with vdW radii
  double* epart;      // [2*nh] cavity energies E1,E2 per subtree
  SubtreeHeader* hdr;  // [nh]
  ushort4* node_pool;
  int pool_cap;
  int* atom_pool;
  int atom_pool_cap;
  int* status;  // [kStatWords]
  char* scratch;  // GLOBAL variant: per-workgroup slab
  size_t scratch_stride;
};

// ---- LDS / scratch carve-out -----------------------------------------------------------------
template <int NCAP, int ACAP>
struct TreeStore {
  double* nd[7];   // node slots: 0-2 centre, 3 exponent, 4 unswitched volume, 5 gamma_1..i, 6 spare
                   // after the bottom-up sweep of a node: 0 psi', 1 E, 2 F_E, 3-5 P_E
  double* at[10];  // local atoms: 0-2 centre, 3 exponent, 4 volume, 5 gamma, 6-8 gradient acc, 9 self-volume acc
  double* cand_vol;
  double* misc;    // [4]: 0 energy accumulator
  int* cand_idx;
  int* at_gidx;
  int* lvl;        // [12]
  unsigned short *nla, *npar, *ncs, *ncc;

  static constexpr size_t kBytes = sizeof(double) * (7 * (size_t)NCAP + 10 * (size_t)ACAP + ACAP + 4) +
                                   sizeof(int) * (2 * (size_t)ACAP + 12) + sizeof(unsigned short) * 4 * (size_t)NCAP;

  __device__ __forceinline__ void carve(char* base) {
    double* d = reinterpret_cast<double*>(base);
    for (int k = 0; k < 7; k++) nd[k] = d + (size_t)k * NCAP;
    d += 7 * (size_t)NCAP;
    for (int k = 0; k < 10; k++) at[k] = d + (size_t)k * ACAP;
    d += 10 * (size_t)ACAP;
    cand_vol = d;
    d += ACAP;
    misc = d;
    d += 4;
    int* ip = reinterpret_cast<int*>(d);
    cand_idx = ip;
    ip += ACAP;
    at_gidx = ip;
    ip += ACAP;
    lvl = ip;
    ip += 12;
    unsigned short* sp = reinterpret_cast<unsigned short*>(ip);
    nla = sp;
    npar = sp + NCAP;
    ncs = sp + 2 * (size_t)NCAP;
    ncc = sp + 3 * (size_t)NCAP;
  }
};

// ---- Gaussian merge ----------------------------------------------------------------------------
struct Merged {
  double x, y, z, a, v;  // overlap Gaussian (v = UNswitched volume)
  double vol;            // switched volume s*v
  double sfp;            // s' * v + s
  double dvx, dvy, dvz;  // dv1 = (c2-c1) * (-dVdr)
  double dvv1;           // dV/dV1 (unswitched)
};

__device__ __forceinline__ double dev_switch(double gvol, double& sp) {
  // gaussvol.cpp:18-41 with volmina/volminb = VOLMINA/VOLMINB
  const double va = kVolMinA, vb = kVolMinB;
  const double w = 1.0 / (vb - va);
  double u = (gvol - va) * w;
  double u2 = u * u;
  double u3 = u * u2;
  double s = u3 * (10.0 - 15.0 * u + 6.0 * u2);
  double d = w * 30.0 * u2 * (1.0 - 2.0 * u + u2);
  if (gvol > vb) {
    s = 1.0;
    d = 0.0;
  } else if (gvol < va) {
    s = 0.0;
    d = 0.0;
  }
  sp = d;
  return s;
}

__device__ __forceinline__ void dev_merge(double x1, double y1, double z1, double a1, double v1, double x2, double y2,
                                          double z2, double a2, double v2, Merged& m) {
  const double dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  const double d2 = dx * dx + dy * dy + dz * dz;
  const double a12 = a1 + a2;
  const double deltai = 1.0 / a12;
  const double df = a1 * a2 * deltai;
  const double ef = exp(-df * d2);
  const double q = df * (1.0 / kPi);
  const double gvol = (v1 * v2) * (q * sqrt(q)) * ef;
  const double mdVdr = 2.0 * df * gvol;  // -(dV/dr)/r
  m.x = (x1 * a1 + x2 * a2) * deltai;
  m.y = (y1 * a1 + y2 * a2) * deltai;
  m.z = (z1 * a1 + z2 * a2) * deltai;
  m.a = a12;
  m.v = gvol;
  double sp;
  const double s = dev_switch(gvol, sp);
  m.vol = s * gvol;
  m.sfp = sp * gvol + s;
  m.dvx = dx * mdVdr;
  m.dvy = dy * mdVdr;
  m.dvz = dz * mdVdr;
  m.dvv1 = v1 > 0 ? gvol / v1 : 0.0;
}

// ---- wave helpers --------------------------------------------------------------------------------
__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ double readlane_f64(double v, int k) {
  unsigned long long u = __double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, k);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void lds_add(double* p, double v) {
  // LDS / global FP64 add; with -munsafe-fp-atomics this is ds_add_f64 / global_atomic_add_f64
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum BuildResult { kBuildOk = 0, kBuildNodeOverflow = 1, kBuildAtomOverflow = 2 };

// ---- rank + append a candidate list held in S.cand_vol / S.cand_idx --------------------------------
// FROM_GLOBAL: candidates are heavy-atom indices (level-2 search); they also become local atoms.
// otherwise  : candidates are sibling node slots.
template <int NCAP, int ACAP, bool FROM_GLOBAL>
__device__ __forceinline__ void append_ranked(const TreeStore<NCAP, ACAP>& S, const TreeArgs& A, int lane, int head, int ncand,
                                              int tail, double x1, double y1, double z1, double a1, double v1, double gam1i) {
  for (int c = lane; c < ncand; c += 64) {
    const double my = S.cand_vol[c];
    int rank = 0;
    for (int k = 0; k < ncand; k++) {
      const double vk = S.cand_vol[k];
      rank += (vk > my || (vk == my && k < c)) ? 1 : 0;
    }
    const int slot = tail + rank;
    int la;
    double x2, y2, z2, a2, v2, g2;
    if (FROM_GLOBAL) {
      const int hj = S.cand_idx[c];
      la = slot;  // level-2 node k <-> local atom k
      x2 = A.hx[hj];
      y2 = A.hy[hj];
      z2 = A.hz[hj];
      a2 = A.a_large[hj];
      v2 = A.v_large[hj];
      g2 = A.gam[hj];
      S.at[0][la] = x2;
      S.at[1][la] = y2;
      S.at[2][la] = z2;
      S.at[3][la] = a2;
      S.at[4][la] = v2;
      S.at[5][la] = g2;
      S.at_gidx[la] = hj;
    } else {
      la = S.nla[S.cand_idx[c]];
      x2 = S.at[0][la];
      y2 = S.at[1][la];
      z2 = S.at[2][la];
      a2 = S.at[3][la];
      v2 = S.at[4][la];
      g2 = S.at[5][la];
    }
    Merged m;
    dev_merge(x1, y1, z1, a1, v1, x2, y2, z2, a2, v2, m);
    S.nd[0][slot] = m.x;
    S.nd[1][slot] = m.y;
    S.nd[2][slot] = m.z;
    S.nd[3][slot] = m.a;
    S.nd[4][slot] = m.v;
    S.nd[5][slot] = gam1i + g2;
    S.nla[slot] = (unsigned short)la;
    S.npar[slot] = (unsigned short)head;
    S.ncs[slot] = 0;
    S.ncc[slot] = 0;
  }
}

// ---- build the subtree of heavy atom `hi` (large radii) ---------------------------------------------
// returns BuildResult; on success *nnodes_out / *natoms_out are set (wave-uniform)
template <int NCAP, int ACAP>
__device__ int build_subtree(const TreeStore<NCAP, ACAP>& S, const TreeArgs& A, int lane, int hi, int* nnodes_out,
                             int* natoms_out) {
  const double rx = A.hx[hi], ry = A.hy[hi], rz = A.hz[hi];
  const double ra = A.a_large[hi], rv = A.v_large[hi], rg = A.gam[hi];
  if (lane == 0) {
    S.at[0][0] = rx;
    S.at[1][0] = ry;
    S.at[2][0] = rz;
    S.at[3][0] = ra;
    S.at[4][0] = rv;
    S.at[5][0] = rg;
    S.at_gidx[0] = hi;
    S.nd[0][0] = rx;
    S.nd[1][0] = ry;
    S.nd[2][0] = rz;
    S.nd[3][0] = ra;
    S.nd[4][0] = rv;
    S.nd[5][0] = rg;
    S.nla[0] = 0;
    S.npar[0] = 0xFFFF;
    S.ncs[0] = 1;
    S.ncc[0] = 0;
    S.misc[0] = 0.0;
  }
  // ---- level 2: all heavy atoms with a larger index whose overlap with the root survives the switch
  int ncand = 0;
  for (int base = hi + 1; base < A.nh; base += 64) {
    const int hj = base + lane;
    bool keep = false;
    double sv = 0.0;
    if (hj < A.nh) {
      const double xj = A.hx[hj], yj = A.hy[hj], zj = A.hz[hj];
      const double dx = xj - rx, dy = yj - ry, dz = zj - rz;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < A.rcut2) {
        Merged m;
        dev_merge(rx, ry, rz, ra, rv, xj, yj, zj, A.a_large[hj], A.v_large[hj], m);
        sv = m.vol;
        keep = sv > kMinGvol;
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      const int cnt = __popcll(mask);
      if (ncand + cnt > ACAP - 1) return kBuildAtomOverflow;
      if (keep) {
        const int p = ncand + lane_prefix(mask);
        S.cand_vol[p] = sv;
        S.cand_idx[p] = hj;
      }
      ncand += cnt;
    }
  }
  if (1 + ncand > NCAP) return kBuildNodeOverflow;
  __syncthreads();
  append_ranked<NCAP, ACAP, true>(S, A, lane, 0, ncand, 1, rx, ry, rz, ra, rv, rg);
  if (lane == 0) {
    S.ncc[0] = (unsigned short)ncand;
    S.lvl[1] = 0;
    S.lvl[2] = 1;
    S.lvl[3] = 1 + ncand;
  }
  __syncthreads();

  // ---- levels 3..8: breadth-first expansion, one head node at a time
  int tail = 1 + ncand;
  int cur = 2;            // level of the head node
  int lvl_next = tail;    // first node of level cur+1
  for (int head = 1; head < tail; ++head) {
    if (head == lvl_next) {
      cur++;
      lvl_next = tail;
      if (lane == 0) S.lvl[cur + 1] = tail;
      if (cur >= kMaxOrder) break;
    }
    const int par = S.npar[head];
    const int sib_end = (int)S.ncs[par] + (int)S.ncc[par];
    const int nsib = sib_end - head - 1;
    if (nsib <= 0) continue;
    const double x1 = S.nd[0][head], y1 = S.nd[1][head], z1 = S.nd[2][head];
    const double a1 = S.nd[3][head], v1 = S.nd[4][head], g1 = S.nd[5][head];
    if (nsib <= 64) {
      // fast path: one candidate per lane, rank by register broadcasts
      const bool valid = lane < nsib;
      const int la = valid ? (int)S.nla[head + 1 + lane] : 0;
      Merged m;
      dev_merge(x1, y1, z1, a1, v1, S.at[0][la], S.at[1][la], S.at[2][la], S.at[3][la], S.at[4][la], m);
      const bool keep = valid && (m.vol > kMinGvol);
      const unsigned long long mask = __ballot(keep);
      if (mask == 0) continue;
      const int cnt = __popcll(mask);
      if (tail + cnt > NCAP) return kBuildNodeOverflow;
      int rank = 0;
      for (unsigned long long bits = mask; bits; bits &= bits - 1) {
        const int k = __builtin_ctzll(bits);
        const double vk = readlane_f64(m.vol, k);
        rank += (vk > m.vol || (vk == m.vol && k < lane)) ? 1 : 0;
      }
      if (keep) {
        const int slot = tail + rank;
        S.nd[0][slot] = m.x;
        S.nd[1][slot] = m.y;
        S.nd[2][slot] = m.z;
        S.nd[3][slot] = m.a;
        S.nd[4][slot] = m.v;
        S.nd[5][slot] = g1 + S.at[5][la];
        S.nla[slot] = (unsigned short)la;
        S.npar[slot] = (unsigned short)head;
        S.ncs[slot] = 0;
        S.ncc[slot] = 0;
      }
      if (lane == 0) {
        S.ncs[head] = (unsigned short)tail;
        S.ncc[head] = (unsigned short)cnt;
      }
      tail += cnt;
      __syncthreads();
    } else {
      // generic path (only reachable when ACAP > 64): chunked scan into the candidate list
      int nc = 0;
      for (int base = head + 1; base < sib_end; base += 64) {
        const int sj = base + lane;
        bool keep = false;
        double sv = 0.0;
        if (sj < sib_end) {
          const int la = S.nla[sj];
          Merged m;
          dev_merge(x1, y1, z1, a1, v1, S.at[0][la], S.at[1][la], S.at[2][la], S.at[3][la], S.at[4][la], m);
          sv = m.vol;
          keep = sv > kMinGvol;
        }
        const unsigned long long mask = __ballot(keep);
        if (mask) {
          if (keep) {
            const int p = nc + lane_prefix(mask);
            S.cand_vol[p] = sv;
            S.cand_idx[p] = sj;
          }
          nc += __popcll(mask);
        }
      }
      if (nc == 0) continue;
      if (tail + nc > NCAP) return kBuildNodeOverflow;
      __syncthreads();
      append_ranked<NCAP, ACAP, false>(S, A, lane, head, nc, tail, x1, y1, z1, a1, v1, g1);
      if (lane == 0) {
        S.ncs[head] = (unsigned short)tail;
        S.ncc[head] = (unsigned short)nc;
      }
      tail += nc;
      __syncthreads();
    }
  }
  if (lane == 0) {
    // lvl[cur+1] was set on entry to level `cur` and equals tail here; deeper levels are empty
    for (int L = cur + 2; L <= 9; L++) S.lvl[L] = tail;
  }
  __syncthreads();
  *nnodes_out = tail;
  *natoms_out = 1 + ncand;
  return kBuildOk;
}

// ---- top-down recompute of every node's Gaussian and gamma from the local atom table ------------------
template <int NCAP, int ACAP>
__device__ void rescan_topdown(const TreeStore<NCAP, ACAP>& S, int lane) {
  if (lane == 0) {
    for (int k = 0; k < 6; k++) S.nd[k][0] = S.at[k][0];
  }
  __syncthreads();
  for (int L = 2; L <= kMaxOrder; L++) {
    const int b = S.lvl[L], e = S.lvl[L + 1];
    if (b >= e) break;
    for (int n = b + lane; n < e; n += 64) {
      const int p = S.npar[n];
      const int la = S.nla[n];
      Merged m;
      dev_merge(S.nd[0][p], S.nd[1][p], S.nd[2][p], S.nd[3][p], S.nd[4][p], S.at[0][la], S.at[1][la], S.at[2][la],
                S.at[3][la], S.at[4][la], m);
      S.nd[0][n] = m.x;
      S.nd[1][n] = m.y;
      S.nd[2][n] = m.z;
      S.nd[3][n] = m.a;
      S.nd[4][n] = m.v;
      S.nd[5][n] = S.nd[5][p] + S.at[5][la];
    }
    __syncthreads();
  }
}

// ---- bottom-up sweep (gaussvol.cpp:400-487) -------------------------------------------------------------
// Accumulates into the local atom table: gradient (at[6..8]) and, if WITH_VOL, self volume (at[9]).
// Returns the subtree energy (valid on every lane) if WITH_VOL.
template <int NCAP, int ACAP, bool WITH_VOL>
__device__ double sweep_bottomup(const TreeStore<NCAP, ACAP>& S, int lane) {
  if (lane == 0) S.misc[0] = 0.0;
  int deepest = 1;
  for (int L = 2; L <= kMaxOrder; L++)
    if (S.lvl[L + 1] > S.lvl[L]) deepest = L;
  __syncthreads();
  for (int L = deepest; L >= 2; --L) {
    const int b = S.lvl[L], e = S.lvl[L + 1];
    const double cf = (L & 1) ? 1.0 : -1.0;
    const double cp = cf / (double)L;
    for (int n = b + lane; n < e; n += 64) {
      const int p = S.npar[n];
      const int la = S.nla[n];
      const double a1 = S.nd[3][p], ai = S.at[3][la];
      Merged m;
      dev_merge(S.nd[0][p], S.nd[1][p], S.nd[2][p], a1, S.nd[4][p], S.at[0][la], S.at[1][la], S.at[2][la], ai, S.at[4][la], m);
      const double gam = S.nd[5][n];
      double psip = cp * m.vol;
      double en = cp * gam * m.vol;
      double fe = cp * m.sfp * gam;
      double pex = 0.0, pey = 0.0, pez = 0.0;
      const int cs = S.ncs[n], cc = S.ncc[n];
      for (int c = cs; c < cs + cc; c++) {
        if (WITH_VOL) {
          psip += S.nd[0][c];
          en += S.nd[1][c];
        }
        fe += S.nd[2][c];
        pex += S.nd[3][c];
        pey += S.nd[4][c];
        pez += S.nd[5][c];
      }
      const double inv_a1i = 1.0 / (a1 + ai);
      const double c2 = ai * inv_a1i;
      lds_add(&S.at[6][la], -m.dvx * fe + pex * c2);
      lds_add(&S.at[7][la], -m.dvy * fe + pey * c2);
      lds_add(&S.at[8][la], -m.dvz * fe + pez * c2);
      if (WITH_VOL) lds_add(&S.at[9][la], psip);
      const double c2p = a1 * inv_a1i;
      const double ox = m.dvx * fe + pex * c2p;
      const double oy = m.dvy * fe + pey * c2p;
      const double oz = m.dvz * fe + pez * c2p;
      const double of = m.dvv1 * fe;
      if (L == 2) {
        // parent is the level-1 root: a_i/a_1i = 1, dv1 = 0 there, so the root's gradient is the sum of P_E
        lds_add(&S.at[6][0], ox);
        lds_add(&S.at[7][0], oy);
        lds_add(&S.at[8][0], oz);
        if (WITH_VOL) {
          lds_add(&S.at[9][0], psip);
          lds_add(&S.misc[0], en);
        }
      } else {
        if (WITH_VOL) {
          S.nd[0][n] = psip;
          S.nd[1][n] = en;
        }
        S.nd[2][n] = of;
        S.nd[3][n] = ox;
        S.nd[4][n] = oy;
        S.nd[5][n] = oz;
      }
    }
    __syncthreads();
  }
  double energy = 0.0;
  if (WITH_VOL) {
    // level-1 node: volume = V_i, coefficient +1 (gaussvol.cpp:138-141)
    const double vroot = S.at[4][0];
    if (lane == 0) lds_add(&S.at[9][0], vroot);
    energy = S.misc[0] + S.at[5][0] * vroot;
    __syncthreads();
  }
  return energy;
}

}  // namespace agbnp
