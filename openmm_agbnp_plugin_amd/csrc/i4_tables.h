// Host-side builder of the AGBNP1 pair-descreening tables Q(r; R_i, R_j).
// Spec: openmmapi/src/AGBNPUtils.cpp:13-130 (integral, switch, nodes) and :134-214 (radius typing);
// spline coefficients by the natural-cubic-spline algorithm of OpenMM's SplineFitter (third party,
// call sites openmmapi/include/AGBNPUtils.h:104-115).
#pragma once
#include <vector>

namespace agbnp {

struct I4TableSet {
  int nscreened = 0;  // distinct radii over all atoms      (table row)
  int nscreener = 0;  // distinct radii over heavy atoms    (table column)
  std::vector<double> radius_screened, radius_screener;
  std::vector<int> type_screened;  // per atom
  std::vector<int> type_screener;  // per atom, -1 for hydrogens
  std::vector<double> y, y2;       // [nscreened*nscreener][16]
  void build(const std::vector<double>& vdw_radius, const std::vector<int>& ishydrogen);
};

}  // namespace agbnp
