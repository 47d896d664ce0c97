#include "i4_tables.h"

#include <cmath>
#include <map>

#include "agbnp_common.h"

namespace agbnp {
namespace {

// Q4 integral of the solvent-excluded sphere j (radius Rj) seen from atom i (radius Ri) at distance r.
// Three geometric regimes (AGBNPUtils.cpp:34-85): separated, partially overlapping, j swallowing i.
double q4_integral(double r, double Ri, double Rj) {
  const double two_pi = 2.0 * M_PI;
  if (r > Ri + Rj) {
    const double rp = r + Rj, rm = r - Rj;
    return two_pi * (Rj / (rp * rm) - 0.5 * log(rp / rm) / r);
  }
  const double gap = Rj - Ri;
  if (r * r > gap * gap) {
    const double rp = r + Rj, rm = r - Rj;
    const double inv_rp = 1. / rp, inv_ri = 1. / Ri;
    const double lg = 0.5 * log(rp / Ri);
    return two_pi * (-(inv_rp - inv_ri) + (0.25 * (rp * rm) * (inv_rp * inv_rp - inv_ri * inv_ri) - lg) / r);
  }
  if (Ri > Rj) return 0.0;
  const double rp = r + Rj, rm = Rj - r;
  const double neg = -rp * rm;  // r^2 - Rj^2
  if (r < .001 * Rj) {          // removable singularity of (1/2a) log((1+a)/(1-a)) at a = 0
    const double a = r / Rj;
    return two_pi * (2. / Ri + Rj / neg - (1. + (2.0 / 3.0) * a * a) / Rj);
  }
  return two_pi * (2. / Ri + Rj / neg - 0.5 * log(rp / rm) / r);
}

// C2 switch from 1 at xa to 0 at xb (AGBNPUtils.cpp:13-25)
double taper(double x, double xa, double xb) {
  if (x > xb) return 0.0;
  if (x < xa) return 1.0;
  const double u = (x - xa) / (xb - xa);
  const double u3 = u * u * u;
  return 1. - u3 * (10. - 15. * u + 6. * u * u);
}

// natural cubic spline second derivatives (tridiagonal system, y''=0 at both ends)
void natural_spline_y2(const std::vector<double>& x, const std::vector<double>& y, double* y2) {
  const int n = (int)x.size();
  std::vector<double> sub(n, 0.0), diag(n, 1.0), sup(n, 0.0), rhs(n, 0.0), ratio(n, 0.0);
  for (int i = 1; i < n - 1; i++) {
    sub[i] = x[i] - x[i - 1];
    diag[i] = 2.0 * (x[i + 1] - x[i - 1]);
    sup[i] = x[i + 1] - x[i];
    rhs[i] = 6.0 * ((y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]));
  }
  y2[0] = rhs[0] / diag[0];
  double pivot = diag[0];
  for (int i = 1; i < n; i++) {
    ratio[i] = sup[i - 1] / pivot;
    pivot = diag[i] - sub[i] * ratio[i];
    y2[i] = (rhs[i] - sub[i] * y2[i - 1]) / pivot;
  }
  for (int i = n - 2; i >= 0; i--) y2[i] -= ratio[i + 1] * y2[i + 1];
}

}  // namespace

void I4TableSet::build(const std::vector<double>& vdw_radius, const std::vector<int>& ishydrogen) {
  const int n = (int)vdw_radius.size();
  // Radius classes: two radii are the same class if long(r*10000) agrees; the class keeps the FIRST
  // radius seen with that key and classes are ordered by key (AGBNPUtils.h:173-179, std::set semantics).
  std::map<long, double> cls_all, cls_heavy;
  for (int i = 0; i < n; i++) cls_all.emplace((long)(vdw_radius[i] * kRadiusPrecision), vdw_radius[i]);
  for (int i = 0; i < n; i++)
    if (!ishydrogen[i]) cls_heavy.emplace((long)((vdw_radius[i] + 0.0) * kRadiusPrecision), vdw_radius[i] + 0.0);
  std::map<long, int> idx_all, idx_heavy;
  radius_screened.clear();
  radius_screener.clear();
  for (auto& kv : cls_all) {
    idx_all[kv.first] = (int)radius_screened.size();
    radius_screened.push_back(kv.second);
  }
  for (auto& kv : cls_heavy) {
    idx_heavy[kv.first] = (int)radius_screener.size();
    radius_screener.push_back(kv.second);
  }
  nscreened = (int)radius_screened.size();
  nscreener = (int)radius_screener.size();
  type_screened.assign(n, -1);
  type_screener.assign(n, -1);
  for (int i = 0; i < n; i++) {
    type_screened[i] = idx_all[(long)(vdw_radius[i] * kRadiusPrecision)];
    if (!ishydrogen[i]) type_screener[i] = idx_heavy[(long)((vdw_radius[i] + 0.0) * kRadiusPrecision)];
  }

  const double rmin = 0.0, rmax = kI4MaxA;
  const double dr = (rmax - rmin) / (kI4Nodes - 1);
  const double xa = 0.5 * (rmax + rmin), xb = rmax;
  std::vector<double> x(kI4Nodes), yy(kI4Nodes);
  for (int k = 0; k < kI4Nodes; k++) x[k] = k * dr + rmin;
  y.assign((size_t)nscreened * nscreener * kI4Nodes, 0.0);
  y2.assign((size_t)nscreened * nscreener * kI4Nodes, 0.0);
  for (int ti = 0; ti < nscreened; ti++)
    for (int tj = 0; tj < nscreener; tj++) {
      const double Ri = radius_screened[ti];
      // AGBNP1: the screener keeps its own radius (gvol12_factor = 0 -> pow(1, 1/3) * Rj, AGBNPUtils.cpp:87-96)
      const double Rj = pow(1.0, 1. / 3.) * radius_screener[tj];
      for (int k = 0; k < kI4Nodes; k++) yy[k] = taper(x[k], xa, xb) * q4_integral(x[k], Ri, Rj);
      const size_t o = ((size_t)ti * nscreener + tj) * kI4Nodes;
      for (int k = 0; k < kI4Nodes; k++) y[o + k] = yy[k];
      natural_spline_y2(x, yy, &y2[o]);
    }
}

}  // namespace agbnp
