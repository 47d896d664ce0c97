// =====================================================================================
//  oracle/agbnp_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
//  A single-threaded FP64 CPU restatement of the reference's "Reference platform" path for
//  AGBNPForce version 0 (GaussVol / GVolSA) and version 1 (AGBNP1).  Only tests/,
//  __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
//  product (openmm_agbnp_plugin_amd/) never links, imports or calls it.
//
//  What it restates (paths relative to /root/reference):
//    gaussvol/gaussvol.cpp:18-41      quintic volume switch               -> vol_switch()
//    gaussvol/gaussvol.cpp:60-93      Gaussian overlap (V,a,c)            -> merge_gaussians()
//    gaussvol/gaussvol.cpp:103-151    level-1 initialisation              -> Tree::seed()
//    gaussvol/gaussvol.cpp:154-250    child scan + sort + append          -> Tree::expand()
//    gaussvol/gaussvol.cpp:376-397    depth-first tree construction       -> Tree::build()
//    gaussvol/gaussvol.cpp:254-327    volume rescan on fixed topology     -> Tree::rescan()
//    gaussvol/gaussvol.cpp:330-372    gamma-only rescan                   -> Tree::rescan_gamma()
//    gaussvol/gaussvol.cpp:400-519    bottom-up volume/energy/gradient    -> Tree::sweep()
//    gaussvol/gaussvol.cpp:584-617    GaussVol facade (force = -grad)     -> volume_pass()
//    openmmapi/src/AGBNPUtils.cpp:13-130   I4 descreening integral + table nodes -> i4_*()
//    openmmapi/src/AGBNPUtils.cpp:134-214  radius typing / 2-D table      -> I4Tables
//    platforms/reference/src/ReferenceAGBNPKernels.cpp:41-55    inverse-Born-radius filter
//    platforms/reference/src/ReferenceAGBNPKernels.cpp:58-137   initialize()  -> oracle_create()
//    platforms/reference/src/ReferenceAGBNPKernels.cpp:152-271  executeGVolSA -> run_v0()
//    platforms/reference/src/ReferenceAGBNPKernels.cpp:274-795  executeAGBNP1 -> run_v1()
//    platforms/reference/src/ReferenceAGBNPKernels.cpp:1796-1815 copyParametersToContext
//
//  Third-party arithmetic that is NOT under /root/reference: OpenMM's SplineFitter
//  (createNaturalSpline / evaluateSpline / evaluateSplineDerivative; OpenMM >= 7.2.2 per the
//  reference README.md:31, no version pin in tree).  Call sites: openmmapi/include/AGBNPUtils.h:104,
//  112,115.  Restated here from its published algorithm (natural cubic spline, tridiagonal solve,
//  bisection interval search, Numerical-Recipes cubic form; the same cubic is restated in-tree at
//  platforms/opencl/src/kernels/AGBNPBornRadii.cl:58-73).
//
//  PINNING (see tests/test_oracle_golden.py):
//    * platforms/reference/tests/v0.reference:4-7  (Volume energy 1/2, Energy 872.514,
//      energy after moving atom 121 by +2e-3 nm in y: 872.576, change 0.0615433,
//      change from gradient 0.0619746)
//    * platforms/reference/tests/v1.reference:2-5  (-2476.66, -2476.58, 0.0874992, 0.0886249)
//    * the 13-digit energies SURVEY.md section 8c / BASELINE.md section 3 record from the survey's
//      run of the reference sources (264-atom fixture, OpenCL-convention fixture, trpcage, 1dwc, 2clr).
//  The reference itself is NOT built here: gaussvol.h:34-38 and every other file on the path
//  include OpenMM headers that this image does not have, so it is "unbuildable here" by the
//  round rules (no stand-in headers).  See DESIGN.md section 3.
// =====================================================================================
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>

namespace {

// ---- model constants: float literals promoted to double exactly as the reference's macros ----
// gaussvol/gaussvol.h:46-63, openmmapi/include/AGBNPForce.h:14-33, AGBNPUtils.h:124-126,155
const double kKFC = (2.2269859253f);
const double kPFC = (2.5f);
const double kMinGvol = FLT_MIN;
const int kMaxOrder = 8;
const double kVolMinA = (0.01f * (0.001f));
const double kVolMinB = (0.1f * (0.001f));
const double kRadiusIncrement = (0.5f * (0.1f));
const double kHBRadius = (1.4 * (0.1f));
const double kI4MaxA = 2.0;
const int kI4Nodes = 16;
const long kRadiusPrecision = 10000;

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Gauss {  // g(x) = v (a/pi)^{3/2} exp(-a |x-c|^2)
  double v, a;
  V3 c;
};

// gaussvol.cpp:18-41
double vol_switch(double gvol, double va, double vb, double& sp) {
  double base = 0.0f, on = 1.0f;
  if (gvol > vb) {
    base = 1.0f;
    on = 0.0f;
  } else if (gvol < va) {
    base = 0.0f;
    on = 0.0f;
  }
  double w = 1.f / (vb - va);
  double u = (gvol - va) * w;
  double u2 = u * u;
  double u3 = u * u2;
  double s = base + on * u3 * (10.f - 15.f * u + 6.f * u2);
  sp = on * w * 30.f * u2 * (1.f - 2.f * u + u2);
  return s;
}

// gaussvol.cpp:60-93 ; returns switched volume, g12.v holds the UNswitched volume
double merge_gaussians(const Gauss& g1, const Gauss& g2, Gauss& g12, double& dVdr, double& dVdV, double& sfp) {
  V3 dist = g2.c - g1.c;
  double d2 = dot(dist, dist);
  double a12 = g1.a + g2.a;
  double deltai = 1. / a12;
  double df = (g1.a) * (g2.a) * deltai;
  double ef = exp(-df * d2);
  double gvol = ((g1.v * g2.v) / pow(M_PI / df, 1.5)) * ef;
  double dgvol = -2.f * df * gvol;
  double dgvolv = g1.v > 0 ? gvol / g1.v : 0.0;
  g12.c = ((g1.c * g1.a) + (g2.c * g2.a)) * deltai;
  g12.a = a12;
  g12.v = gvol;
  double sp;
  double s = vol_switch(gvol, kVolMinA, kVolMinB, sp);
  sfp = sp * gvol + s;
  dVdr = dgvol;
  dVdV = dgvolv;
  return s * gvol;
}

struct Node {
  int level;
  Gauss g;
  double volume;  // switched
  double dvv1;
  V3 dv1;
  double gamma1i;
  double sfp;
  int atom, parent, cstart, ccount;
};

struct Sweep {  // subtree accumulators of gaussvol.cpp:400-487
  double psi, f;
  V3 p;
  double psip, fp;
  V3 pp;
  double e, fe;
  V3 pe;
};

struct Tree {
  int natoms = 0;
  std::vector<Node> nodes;  // 0 = root, 1..N = atoms (same layout as gaussvol.h:202)

  void seed_atoms(const std::vector<V3>& pos, const std::vector<double>& radius, const std::vector<double>& volume,
                  const std::vector<double>& gamma, const std::vector<int>& ish, bool fresh) {
    if (fresh) {
      nodes.clear();
      nodes.resize(natoms + 1);
    }
    Node& r = nodes[0];
    r.level = 0;
    r.volume = 0;
    r.dv1 = {0, 0, 0};
    r.dvv1 = 0.;
    r.sfp = 1.;
    r.gamma1i = 0.;
    if (fresh) {
      r.g = Gauss{0, 0, {0, 0, 0}};
      r.parent = -1;
      r.atom = -1;
      r.cstart = 1;
      r.ccount = natoms;
    }
    for (int i = 0; i < natoms; i++) {
      Node& n = nodes[i + 1];
      double a = kKFC / (radius[i] * radius[i]);
      double vol = ish[i] > 0 ? 0. : volume[i];
      n.level = 1;
      n.g.v = vol;
      n.g.a = a;
      n.g.c = pos[i];
      n.volume = vol;
      n.dv1 = {0, 0, 0};
      n.dvv1 = 1.;
      n.sfp = 1.;
      n.gamma1i = gamma[i];
      if (fresh) {
        n.parent = 0;
        n.atom = i;
        n.cstart = -1;
        n.ccount = -1;
      }
    }
  }

  // gaussvol.cpp:197-250 + 154-192: children of `slot` = overlaps with the last atoms of its
  // YOUNGER siblings, kept if switched volume > FLT_MIN, sorted by switched volume (descending),
  // appended at the tail.  Returns first child slot, n in *count.
  int expand(int slot, int* count) {
    *count = 0;
    int parent = nodes[slot].parent;
    if (parent < 0 || nodes[slot].level >= kMaxOrder) return -1;
    int sib_end = nodes[parent].cstart + nodes[parent].ccount;
    std::vector<Node> kids;
    for (int sj = slot + 1; sj < sib_end; sj++) {
      int atom2 = nodes[sj].atom;
      const Gauss& g1 = nodes[slot].g;
      const Gauss& g2 = nodes[atom2 + 1].g;
      Gauss g12;
      double dVdr, dVdV, sfp;
      double gvol = merge_gaussians(g1, g2, g12, dVdr, dVdV, sfp);
      if (gvol > kMinGvol) {
        Node k;
        k.g = g12;
        k.volume = gvol;
        k.atom = atom2;
        k.dv1 = (g2.c - g1.c) * (-dVdr);
        k.dvv1 = dVdV;
        k.sfp = sfp;
        k.gamma1i = nodes[slot].gamma1i + nodes[atom2 + 1].gamma1i;
        kids.push_back(k);
      }
    }
    if (kids.empty()) return -1;
    std::sort(kids.begin(), kids.end(), [](const Node& a, const Node& b) { return a.volume > b.volume; });
    int start = (int)nodes.size();
    int lvl = nodes[slot].level + 1;
    nodes[slot].cstart = start;
    nodes[slot].ccount = (int)kids.size();
    for (Node& k : kids) {
      k.level = lvl;
      k.parent = slot;
      k.cstart = -1;
      k.ccount = -1;
      nodes.push_back(k);
    }
    *count = (int)kids.size();
    return start;
  }

  void grow(int slot) {  // gaussvol.cpp:376-387 (depth first, children appended before descending)
    int n;
    int start = expand(slot, &n);
    for (int k = 0; k < n; k++) grow(start + k);
  }

  void build(const std::vector<V3>& pos, const std::vector<double>& radius, const std::vector<double>& volume,
             const std::vector<double>& gamma, const std::vector<int>& ish) {
    seed_atoms(pos, radius, volume, gamma, ish, true);
    for (int s = 1; s <= natoms; s++) grow(s);
  }

  void rescan_node(int slot) {  // gaussvol.cpp:254-287
    Node& ov = nodes[slot];
    if (ov.parent > 0) {
      const Gauss& g1 = nodes[ov.parent].g;
      const Gauss& g2 = nodes[ov.atom + 1].g;
      Gauss g12;
      double dVdr, dVdV, sfp;
      double gvol = merge_gaussians(g1, g2, g12, dVdr, dVdV, sfp);
      ov.g = g12;
      ov.volume = gvol;
      ov.dv1 = (g2.c - g1.c) * (-dVdr);
      ov.dvv1 = dVdV;
      ov.sfp = sfp;
      ov.gamma1i = nodes[ov.parent].gamma1i + nodes[ov.atom + 1].gamma1i;
    }
    for (int c = nodes[slot].cstart; c < nodes[slot].cstart + nodes[slot].ccount; c++) rescan_node(c);
  }

  void rescan(const std::vector<V3>& pos, const std::vector<double>& radius, const std::vector<double>& volume,
              const std::vector<double>& gamma, const std::vector<int>& ish) {
    seed_atoms(pos, radius, volume, gamma, ish, false);
    nodes[0].level = 0;
    rescan_node(0);
  }

  void rescan_gamma_node(int slot) {  // gaussvol.cpp:330-351
    Node& ov = nodes[slot];
    if (ov.parent > 0) ov.gamma1i = nodes[ov.parent].gamma1i + nodes[ov.atom + 1].gamma1i;
    for (int c = ov.cstart; c < ov.cstart + ov.ccount; c++) rescan_gamma_node(c);
  }

  void rescan_gamma(const std::vector<double>& gamma) {  // gaussvol.cpp:356-372
    nodes[0].gamma1i = 0.;
    for (int i = 0; i < natoms; i++) nodes[i + 1].gamma1i = gamma[i];
    rescan_gamma_node(0);
  }

  // gaussvol.cpp:400-487
  void sweep(int slot, Sweep& S, std::vector<V3>& dr, std::vector<double>& dv, std::vector<double>& freev,
             std::vector<double>& selfv) {
    const Node& ov = nodes[slot];
    double cf = ov.level % 2 == 0 ? -1.0 : 1.0;
    double coeff = ov.level > 0 ? cf : 0;
    double coeffp = ov.level > 0 ? coeff / (double)ov.level : 0;
    int atom = ov.atom;
    double ai = nodes[atom + 1].g.a;  // for the root (atom=-1) this reads slot 0, value unused
    double a1i = ov.g.a;
    double a1 = a1i - ai;

    S.psi = coeff * ov.volume;
    S.f = coeff * ov.sfp;
    S.p = {0, 0, 0};
    S.psip = coeffp * ov.volume;
    S.fp = coeffp * ov.sfp;
    S.pp = {0, 0, 0};
    S.e = coeffp * ov.gamma1i * ov.volume;
    S.fe = coeffp * ov.sfp * ov.gamma1i;
    S.pe = {0, 0, 0};

    if (ov.cstart >= 0) {
      for (int c = ov.cstart; c < ov.cstart + ov.ccount; c++) {
        Sweep T;
        sweep(c, T, dr, dv, freev, selfv);
        S.psi += T.psi;
        S.f += T.f;
        S.p = S.p + T.p;
        S.psip += T.psip;
        S.fp += T.fp;
        S.pp = S.pp + T.pp;
        S.e += T.e;
        S.fe += T.fe;
        S.pe = S.pe + T.pe;
      }
    }
    if (ov.level > 0) {
      freev[atom] += S.psi;
      selfv[atom] += S.psip;
      double c2 = ai / a1i;
      dr[atom] = dr[atom] + (neg(ov.dv1) * S.fe + S.pe * c2);
      dv[atom] += ov.g.v * S.fe;
      c2 = a1 / a1i;
      S.p = ov.dv1 * S.f + S.p * c2;
      S.pp = ov.dv1 * S.fp + S.pp * c2;
      S.pe = ov.dv1 * S.fe + S.pe * c2;
      S.f = ov.dvv1 * S.f;
      S.fp = ov.dvv1 * S.fp;
      S.fe = ov.dvv1 * S.fe;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Natural cubic spline (OpenMM SplineFitter, published algorithm; see header).
// ---------------------------------------------------------------------------------------------
void natural_spline(const std::vector<double>& x, const std::vector<double>& y, std::vector<double>& y2) {
  int n = (int)x.size();
  y2.assign(n, 0.0);
  std::vector<double> a(n), b(n), c(n), rhs(n), gam(n);
  a[0] = 0.0;
  b[0] = 1.0;
  c[0] = 0.0;
  rhs[0] = 0.0;
  for (int i = 1; i < n - 1; i++) {
    a[i] = x[i] - x[i - 1];
    b[i] = 2.0 * (x[i + 1] - x[i - 1]);
    c[i] = x[i + 1] - x[i];
    rhs[i] = 6.0 * ((y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]));
  }
  a[n - 1] = 0.0;
  b[n - 1] = 1.0;
  c[n - 1] = 0.0;
  rhs[n - 1] = 0.0;
  y2[0] = rhs[0] / b[0];
  double beta = b[0];
  for (int i = 1; i < n; i++) {
    gam[i] = c[i - 1] / beta;
    beta = b[i] - a[i] * gam[i];
    y2[i] = (rhs[i] - a[i] * y2[i - 1]) / beta;
  }
  for (int i = n - 2; i >= 0; i--) y2[i] -= gam[i + 1] * y2[i + 1];
}

inline void spline_interval(const std::vector<double>& x, double t, int& lo, int& hi) {
  lo = 0;
  hi = (int)x.size() - 1;
  while (hi - lo > 1) {
    int mid = (hi + lo) / 2;
    if (x[mid] > t)
      hi = mid;
    else
      lo = mid;
  }
}

double spline_value(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& y2, double t) {
  int lo, hi;
  spline_interval(x, t, lo, hi);
  double dx = x[hi] - x[lo];
  double a = (x[hi] - t) / dx;
  double b = 1.0 - a;
  return a * y[lo] + b * y[hi] + ((a * a * a - a) * y2[lo] + (b * b * b - b) * y2[hi]) * dx * dx / 6.0;
}

double spline_deriv(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& y2, double t) {
  int lo, hi;
  spline_interval(x, t, lo, hi);
  double dx = x[hi] - x[lo];
  double a = (x[hi] - t) / dx;
  double b = 1.0 - a;
  double dadx = -1.0 / dx;
  return dadx * y[lo] - dadx * y[hi] + ((1.0 - 3.0 * a * a) * y2[lo] + (3.0 * b * b - 1.0) * y2[hi]) * dx / 6.0;
}

// ---------------------------------------------------------------------------------------------
// I4 descreening integral and its tables (AGBNPUtils.cpp:13-130)
// ---------------------------------------------------------------------------------------------
double i4_switch(double x, double xa, double xb) {
  if (x > xb) return 0.0;
  if (x < xa) return 1.0;
  double d = 1. / (xb - xa);
  double u = (x - xa) * d;
  double u2 = u * u;
  double u3 = u * u2;
  return 1. - u3 * (10. - 15. * u + 6. * u2);
}

double i4_analytic(double rij, double Ri, double Rj) {
  const double twopi = 2.0 * M_PI;
  const double twothirds = 2.0 / 3.0;
  double rij2 = rij * rij;
  double q;
  if (rij > (Ri + Rj)) {
    double u1 = rij + Rj, u2 = rij - Rj;
    double u3 = u1 * u2;
    double u4 = 0.5 * log(u1 / u2);
    q = twopi * (Rj / u3 - u4 / rij);
  } else {
    double u1 = Rj - Ri;
    if (rij2 > u1 * u1) {  // partial overlap
      u1 = rij + Rj;
      double u2 = rij - Rj;
      double u3 = u1 * u2;
      double u4 = 1. / u1;
      double u4sq = u4 * u4;
      double u5 = 1. / Ri;
      double u5sq = u5 * u5;
      double u6 = 0.5 * log(u1 / Ri);
      q = twopi * (-(u4 - u5) + (0.25 * u3 * (u4sq - u5sq) - u6) / rij);
    } else {  // inclusion
      if (Ri > Rj) {
        q = 0.0;
      } else {
        u1 = rij + Rj;
        double u2 = Rj - rij;
        double u3 = -u1 * u2;
        if (rij < .001 * Rj) {
          double a = rij / Rj;
          double u6 = (1. + twothirds * a * a) / Rj;
          q = twopi * (2. / Ri + Rj / u3 - u6);
        } else {
          double u6 = 0.5 * log(u1 / u2);
          q = twopi * (2. / Ri + Rj / u3 - u6 / (rij));
        }
      }
    }
  }
  return q;
}

// AGBNPUtils.cpp:27-32,87-96 with gvol12_factor (0 for AGBNP1 -> newRj == Rj up to pow(1,1/3))
double i4_with_overlap(double rij, double Ri, double Rj, double gvol12_factor) {
  double ai = kKFC / (Ri * Ri);
  double aj = kKFC / (Rj * Rj);
  double d2 = rij * rij;
  double deltai = 1. / (ai + aj);
  double gvol = kPFC * kPFC * exp(-ai * aj * d2 * deltai) * pow(M_PI * deltai, 1.5);
  double volj = 4. * M_PI * Rj * Rj * Rj / 3.;
  double newRj = pow((volj + gvol12_factor * gvol) / volj, 1. / 3.) * Rj;
  return i4_analytic(rij, Ri, newRj);
}

struct I4Tables {
  int nscreened = 0, nscreener = 0, nnodes = kI4Nodes;
  std::vector<double> x;              // nodes (shared)
  std::vector<std::vector<double>> y, y2;  // [ti*nscreener+tj][node]
  std::vector<int> type_screened, type_screener;
  std::vector<double> R_screened, R_screener;

  struct LessTrunc {  // AGBNPUtils.h:173-179
    bool operator()(const double& l, const double& r) const {
      long il = l * kRadiusPrecision;
      long ir = r * kRadiusPrecision;
      return il < ir;
    }
  };

  void build(const std::vector<double>& radii, const std::vector<int>& ish, double rmin, double rmax) {
    std::set<double, LessTrunc> si, sj;
    for (size_t i = 0; i < radii.size(); i++) si.insert(radii[i]);
    double roff = 0.0;
    for (size_t i = 0; i < radii.size(); i++)
      if (!ish[i]) sj.insert(radii[i] + roff);
    nscreened = (int)si.size();
    nscreener = (int)sj.size();
    R_screened.assign(si.begin(), si.end());
    R_screener.assign(sj.begin(), sj.end());
    double dr = (rmax - rmin) / (nnodes - 1);
    double xa = 0.5 * (rmax + rmin), xb = rmax;
    x.resize(nnodes);
    for (int k = 0; k < nnodes; k++) x[k] = k * dr + rmin;
    y.assign((size_t)nscreened * nscreener, std::vector<double>());
    y2.assign((size_t)nscreened * nscreener, std::vector<double>());
    for (int ti = 0; ti < nscreened; ti++)
      for (int tj = 0; tj < nscreener; tj++) {
        std::vector<double> yy(nnodes);
        for (int k = 0; k < nnodes; k++) yy[k] = i4_switch(x[k], xa, xb) * i4_with_overlap(x[k], R_screened[ti], R_screener[tj], 0.0);
        size_t idx = (size_t)ti * nscreener + tj;
        y[idx] = yy;
        natural_spline(x, yy, y2[idx]);
      }
    type_screened.assign(radii.size(), -1);
    type_screener.assign(radii.size(), -1);
    for (size_t i = 0; i < radii.size(); i++) {
      auto it = si.find(radii[i]);
      type_screened[i] = (int)std::distance(si.begin(), it);
      if (!ish[i]) {
        auto jt = sj.find(radii[i] + roff);
        type_screener[i] = (int)std::distance(sj.begin(), jt);
      }
    }
  }
  double eval(double d, int ti, int tj) const { return spline_value(x, y[(size_t)ti * nscreener + tj], y2[(size_t)ti * nscreener + tj], d); }
  double evalderiv(double d, int ti, int tj) const { return spline_deriv(x, y[(size_t)ti * nscreener + tj], y2[(size_t)ti * nscreener + tj], d); }
};

// ReferenceAGBNPKernels.cpp:41-55
double invbr_filter(double beta, double& fp) {
  const double a = 1. / kI4MaxA;
  const double a2 = 1. / (kI4MaxA * kI4MaxA);
  double t;
  if (beta < 0.0) {
    t = a;
    fp = 0.0;
  } else {
    t = sqrt(a2 + beta * beta);
    fp = beta / t;
  }
  return t;
}

struct Oracle {
  int n = 0, version = 1;
  double roffset = kRadiusIncrement;
  std::vector<double> r_large, r_vdw, gammas, alpha, charge;
  std::vector<int> ish;
  Tree tree;
  I4Tables lut;
  std::string err;
  // diagnostics of the last execute()
  double e_vol1 = 0, e_vol2 = 0, e_gb = 0, e_vdw = 0, volume1 = 0, volume2 = 0;
  std::vector<double> selfvol_large, selfvol_vdw, freevol_vdw, born, scale, brw, bru, Y, W, U;
  std::vector<int> level_counts;
  long nslots_build = 0;

  // GaussVol::compute_volume (gaussvol.cpp:588-605)
  void volume_pass(const std::vector<V3>& /*pos*/, const std::vector<double>& volumes, double& volume, double& energy,
                   std::vector<V3>& force, std::vector<double>& gradV, std::vector<double>& freev, std::vector<double>& selfv) {
    for (auto& f : force) f = {0, 0, 0};
    std::fill(gradV.begin(), gradV.end(), 0.);
    std::fill(freev.begin(), freev.end(), 0.);
    std::fill(selfv.begin(), selfv.end(), 0.);
    Sweep S;
    tree.sweep(0, S, force, gradV, freev, selfv);
    volume = S.psi;
    energy = S.e;
    for (int i = 0; i < n; i++) force[i] = neg(force[i]);
    for (int i = 0; i < n; i++)
      if (volumes[i] > 0) gradV[i] = gradV[i] / volumes[i];
  }

  double run_cavity(const std::vector<V3>& pos, std::vector<V3>& force, std::vector<double>& selfv_out) {
    std::vector<double> nu(n), vol_large(n), vol_vdw(n), gradV(n), freev(n), selfv(n);
    std::vector<V3> vforce(n);
    for (int i = 0; i < n; i++) nu[i] = gammas[i] / roffset;
    for (int i = 0; i < n; i++) vol_large[i] = ish[i] > 0 ? 0.0 : 4. * M_PI * pow(r_large[i], 3) / 3.;
    tree.build(pos, r_large, vol_large, nu, ish);
    nslots_build = (long)tree.nodes.size();
    level_counts.assign(kMaxOrder + 1, 0);
    for (const Node& nd : tree.nodes) level_counts[nd.level]++;
    volume_pass(pos, vol_large, volume1, e_vol1, vforce, gradV, freev, selfv);
    selfvol_large = selfv;
    for (int i = 0; i < n; i++) force[i] = force[i] + vforce[i] * 1.0;
    double energy = e_vol1 * 1.0;

    for (int i = 0; i < n; i++) nu[i] = -gammas[i] / roffset;
    for (int i = 0; i < n; i++) vol_vdw[i] = ish[i] > 0 ? 0.0 : 4. * M_PI * pow(r_vdw[i], 3) / 3.;
    tree.rescan(pos, r_vdw, vol_vdw, nu, ish);
    volume_pass(pos, vol_vdw, volume2, e_vol2, vforce, gradV, freev, selfv);
    for (int i = 0; i < n; i++) force[i] = force[i] + vforce[i] * 1.0;
    energy += e_vol2 * 1.0;
    selfvol_vdw = selfv;
    freevol_vdw = freev;
    selfv_out = selfv;
    return energy;
  }

  double run_v0(const std::vector<V3>& pos, std::vector<V3>& force) {
    std::vector<double> sv;
    e_gb = e_vdw = 0;
    return run_cavity(pos, force, sv);
  }

  // FAST-MODE switch (not the Reference platform): cutoff2 > 0 restates the pair truncation of the reference's OpenCL
  // platform on top of the same FP64 arithmetic -- every pair loop of AGBNP1 skips pairs with r^2 >= CUTOFF_SQUARED
  // (platforms/opencl/src/kernels/AGBNPBornRadii.cl:268,430 descreening sums and their derivative pass,
  // AGBNPGBEnergy.cl:145,186 GB pairs; all of them inside #ifdef USE_CUTOFF, which that platform only defines for a
  // nonbonded method other than NoCutoff, OpenCLAGBNPKernels.cpp:487,1149-1150 -- the caller of this switch applies that
  // gate: oracle.py passes no cutoff for NoCutoff).  No reference-held vector exists for that platform: this mode is PARITY UNPINNED;
  // it is tied to the pinned path by the limit cutoff -> infinity (tests/test_oracle_golden.py).
  double cutoff2 = 0.0;

  double run_v1(const std::vector<V3>& pos, std::vector<V3>& force) {
    std::vector<double> selfv;
    double energy = run_cavity(pos, force, selfv);

    scale.assign(n, 0.);
    for (int i = 0; i < n; i++) {
      double rad = r_vdw[i];
      double vol = (4. / 3.) * M_PI * rad * rad * rad;
      scale[i] = selfv[i] / vol;
    }
    const double pifac = 1. / (4. * M_PI);
    born.assign(n, 0.);
    std::vector<double> invbr(n), invbr_fp(n);
    for (int i = 0; i < n; i++) {
      invbr[i] = 1. / r_vdw[i];
      for (int j = 0; j < n; j++) {
        if (i == j) continue;
        if (ish[j] > 0) continue;
        V3 dist = pos[j] - pos[i];
        if (cutoff2 > 0.0 && dot(dist, dist) >= cutoff2) continue;
        double d = sqrt(dot(dist, dist));
        if (d < kI4MaxA) invbr[i] -= pifac * scale[j] * lut.eval(d, lut.type_screened[i], lut.type_screener[j]);
      }
      double fp;
      born[i] = 1. / invbr_filter(invbr[i], fp);
      invbr_fp[i] = fp;
    }

    const double tokjmol = 4.184 * 332.0 / 10.0;
    const double diel = tokjmol * (-0.5) * (1. / 1.0 - 1. / 80.0);
    const double pt25 = 0.25;
    Y.assign(n, 0.);
    double gb_self = 0, gb_pair = 0;
    for (int i = 0; i < n; i++) {
      gb_self += diel * charge[i] * charge[i] / born[i];
      for (int j = i + 1; j < n; j++) {
        V3 dist = pos[j] - pos[i];
        double d2 = dot(dist, dist);
        if (cutoff2 > 0.0 && d2 >= cutoff2) continue;
        double qqf = charge[j] * charge[i];
        double qq = diel * qqf;
        double bb = born[i] * born[j];
        double etij = exp(-pt25 * d2 / bb);
        double fgb = 1. / sqrt(d2 + bb * etij);
        gb_pair += 2. * qq * fgb;
        double fgb3 = fgb * fgb * fgb;
        double mw = -2.0 * qq * (1.0 - pt25 * etij) * fgb3;
        V3 g = dist * mw;
        force[i] = force[i] + g;
        force[j] = force[j] - g;
        double ytij = qqf * (bb + pt25 * d2) * etij * fgb3;
        Y[i] += ytij;
        Y[j] += ytij;
      }
    }
    e_gb = gb_pair + gb_self;
    energy += gb_pair + gb_self;

    double evdw = 0.;
    for (int i = 0; i < n; i++) evdw += alpha[i] / pow(born[i] + kHBRadius, 3);
    e_vdw = evdw;
    energy += evdw;

    brw.assign(n, 0.);
    bru.assign(n, 0.);
    for (int i = 0; i < n; i++) {
      double br = born[i];
      brw[i] = -pifac * 3. * alpha[i] * br * br * invbr_fp[i] / pow(br + kHBRadius, 4);
      bru[i] = -pifac * diel * (charge[i] * charge[i] + Y[i] * br) * invbr_fp[i];
    }

    W.assign(n, 0.);
    U.assign(n, 0.);
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) {
        if (i == j) continue;
        if (ish[j] > 0) continue;
        V3 dist = pos[j] - pos[i];
        if (cutoff2 > 0.0 && dot(dist, dist) >= cutoff2) continue;
        double d = sqrt(dot(dist, dist));
        double Q = 0.0, dQ = 0.0;
        if (d < kI4MaxA) {
          Q = lut.eval(d, lut.type_screened[i], lut.type_screener[j]);
          dQ = lut.evalderiv(d, lut.type_screened[i], lut.type_screener[j]);
        }
        // the reference evaluates ((dist*br)*s_j)*dQ then divides the vector by d
        // (OpenMM Vec3::operator/ multiplies by 1/d)
        double invd = 1.0 / d;
        W[j] += brw[i] * Q;
        V3 w = (((dist * brw[i]) * scale[j]) * dQ) * invd;
        force[i] = force[i] + w;
        force[j] = force[j] - w;
        U[j] += bru[i] * Q;
        w = (((dist * bru[i]) * scale[j]) * dQ) * invd;
        force[i] = force[i] + w;
        force[j] = force[j] - w;
      }
    }

    std::vector<double> nu(n), vol_vdw(n), gradV(n), freev(n), sv(n);
    std::vector<V3> vforce(n);
    for (int i = 0; i < n; i++) vol_vdw[i] = ish[i] > 0 ? 0.0 : 4. * M_PI * pow(r_vdw[i], 3) / 3.;
    double vtmp, etmp;
    for (int i = 0; i < n; i++) nu[i] = W[i] / (4. * M_PI * pow(r_vdw[i], 3) / 3.0);
    tree.rescan_gamma(nu);
    volume_pass(pos, vol_vdw, vtmp, etmp, vforce, gradV, freev, sv);
    for (int i = 0; i < n; i++) force[i] = force[i] + vforce[i];
    for (int i = 0; i < n; i++) nu[i] = U[i] / (4. * M_PI * pow(r_vdw[i], 3) / 3.0);
    tree.rescan_gamma(nu);
    volume_pass(pos, vol_vdw, vtmp, etmp, vforce, gradV, freev, sv);
    for (int i = 0; i < n; i++) force[i] = force[i] + vforce[i];
    return energy;
  }
};

}  // namespace

// =====================================================================================
// C ABI for ctypes (tests / smoke / bench cpu_baseline only)
// =====================================================================================
extern "C" {

void* agbnp_oracle_create(int n, const double* radius, const double* gamma, const double* alpha, const double* charge,
                          const int* ishydrogen, int version, char* errbuf, int errlen) {
  Oracle* o = new Oracle();
  o->n = n;
  o->version = version;
  auto fail = [&](const char* msg) -> void* {
    if (errbuf && errlen > 0) {
      strncpy(errbuf, msg, errlen - 1);
      errbuf[errlen - 1] = 0;
    }
    delete o;
    return nullptr;
  };
  if (version < 0 || version > 1) return fail("oracle: only versions 0 (GVolSA) and 1 (AGBNP1) are restated");
  o->roffset = kRadiusIncrement;
  o->r_large.resize(n);
  o->r_vdw.resize(n);
  o->gammas.resize(n);
  o->alpha.resize(n);
  o->charge.resize(n);
  o->ish.resize(n);
  double common_gamma = -1;
  for (int i = 0; i < n; i++) {
    bool h = ishydrogen[i] != 0;
    o->r_large[i] = radius[i] + o->roffset;
    o->r_vdw[i] = radius[i];
    o->gammas[i] = h ? 0.0 : gamma[i];
    o->alpha[i] = alpha[i];
    o->charge[i] = charge[i];
    o->ish[i] = h ? 1 : 0;
    if (common_gamma < 0 && !h) {
      common_gamma = gamma[i];
    } else if (!h && pow(common_gamma - gamma[i], 2) > FLT_MIN) {
      return fail("initialize(): AGBNP does not support multiple gamma values.");
    }
  }
  o->tree.natoms = n;
  o->lut.build(o->r_vdw, o->ish, 0., kI4MaxA);
  return o;
}

void agbnp_oracle_destroy(void* h) { delete (Oracle*)h; }

// fast-mode switch (see Oracle::cutoff2): cutoff <= 0 restores the Reference platform's semantics
void agbnp_oracle_set_cutoff(void* h, double cutoff) { ((Oracle*)h)->cutoff2 = cutoff > 0.0 ? cutoff * cutoff : 0.0; }

// copyParametersToContext semantics; returns 0 ok, -1 error (message in errbuf)
int agbnp_oracle_update(void* h, int n, const double* radius, const double* gamma, const double* alpha, const double* charge,
                        const int* ishydrogen, char* errbuf, int errlen) {
  Oracle* o = (Oracle*)h;
  auto fail = [&](const char* msg) {
    if (errbuf && errlen > 0) {
      strncpy(errbuf, msg, errlen - 1);
      errbuf[errlen - 1] = 0;
    }
    return -1;
  };
  if (n != o->n) return fail("updateParametersInContext: The number of AGBNP particles has changed");
  for (int i = 0; i < n; i++) {
    if (pow(o->r_vdw[i] - radius[i], 2) > 1.e-6)
      return fail("updateParametersInContext: AGBNP plugin does not support changing atomic radii.");
    if (ishydrogen[i] && o->ish[i] == 0)
      return fail("updateParametersInContext: AGBNP plugin does not support changing heavy/hydrogen atoms.");
    o->gammas[i] = ishydrogen[i] ? 0.0 : gamma[i];
    o->alpha[i] = alpha[i];
    o->charge[i] = charge[i];
  }
  return 0;
}

// forces are ACCUMULATED into force[3n] (as the Reference platform does), energy is returned
int agbnp_oracle_execute(void* h, const double* pos, double* force, double* energy) {
  Oracle* o = (Oracle*)h;
  int n = o->n;
  std::vector<V3> p(n), f(n);
  for (int i = 0; i < n; i++) {
    p[i] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    f[i] = {force[3 * i], force[3 * i + 1], force[3 * i + 2]};
  }
  double e = o->version == 0 ? o->run_v0(p, f) : o->run_v1(p, f);
  for (int i = 0; i < n; i++) {
    force[3 * i] = f[i].x;
    force[3 * i + 1] = f[i].y;
    force[3 * i + 2] = f[i].z;
  }
  *energy = e;
  return 0;
}

// scalar diagnostics: 0 E_vol1, 1 E_vol2, 2 E_gb, 3 E_vdw, 4 volume1, 5 volume2, 6 slots, 7 nscreened, 8 nscreener
double agbnp_oracle_scalar(void* h, int which) {
  Oracle* o = (Oracle*)h;
  switch (which) {
    case 0: return o->e_vol1;
    case 1: return o->e_vol2;
    case 2: return o->e_gb;
    case 3: return o->e_vdw;
    case 4: return o->volume1;
    case 5: return o->volume2;
    case 6: return (double)o->nslots_build;
    case 7: return (double)o->lut.nscreened;
    case 8: return (double)o->lut.nscreener;
  }
  return 0;
}

// per-atom diagnostics: 0 self volume (large radii), 1 self volume (vdW radii), 2 Born radius,
// 3 volume scaling factor, 4 brw, 5 bru, 6 Y, 7 W, 8 U, 9 free volume (vdW)
int agbnp_oracle_vector(void* h, int which, double* out) {
  Oracle* o = (Oracle*)h;
  const std::vector<double>* v = nullptr;
  switch (which) {
    case 0: v = &o->selfvol_large; break;
    case 1: v = &o->selfvol_vdw; break;
    case 2: v = &o->born; break;
    case 3: v = &o->scale; break;
    case 4: v = &o->brw; break;
    case 5: v = &o->bru; break;
    case 6: v = &o->Y; break;
    case 7: v = &o->W; break;
    case 8: v = &o->U; break;
    case 9: v = &o->freevol_vdw; break;
  }
  if (!v || (int)v->size() != o->n) return -1;
  memcpy(out, v->data(), sizeof(double) * o->n);
  return 0;
}

// tree statistics of the build pass: counts[0..8] nodes per level, then max subtree size under a
// level-1 atom (excluding the atom) and max children of any node
int agbnp_oracle_tree_stats(void* h, long* counts9, long* max_subtree, long* max_children) {
  Oracle* o = (Oracle*)h;
  if (o->tree.nodes.empty()) return -1;
  for (int l = 0; l <= 8; l++) counts9[l] = o->level_counts[l];
  long mc = 0;
  std::vector<long> under(o->tree.nodes.size(), 0);
  for (long s = (long)o->tree.nodes.size() - 1; s >= 1; s--) {
    const Node& nd = o->tree.nodes[s];
    if (nd.ccount > mc) mc = nd.ccount;
    if (nd.parent > 0) under[nd.parent] += under[s] + 1;
  }
  long ms = 0;
  for (int i = 1; i <= o->n; i++) ms = std::max(ms, under[i]);
  *max_subtree = ms;
  *max_children = mc;
  return 0;
}

// I4 tables: sizes via scalar(7), scalar(8); y,y2 are [nscreened*nscreener][16]; types per atom
int agbnp_oracle_tables(void* h, double* x16, double* y, double* y2, int* type_screened, int* type_screener) {
  Oracle* o = (Oracle*)h;
  int nt = o->lut.nscreened * o->lut.nscreener;
  for (int k = 0; k < kI4Nodes; k++) x16[k] = o->lut.x[k];
  for (int t = 0; t < nt; t++)
    for (int k = 0; k < kI4Nodes; k++) {
      y[t * kI4Nodes + k] = o->lut.y[t][k];
      y2[t * kI4Nodes + k] = o->lut.y2[t][k];
    }
  for (int i = 0; i < o->n; i++) {
    type_screened[i] = o->lut.type_screened[i];
    type_screener[i] = o->lut.type_screener[i];
  }
  return 0;
}

}  // extern "C"
