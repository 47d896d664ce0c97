// Branch-free FP64 building blocks of the gfx950 kernels (device only).
#pragma once
#include <hip/hip_runtime.h>

namespace agbnp {

// 2^y for y <= 0, relative error 4.4e-16 on the reduced interval, no special cases: the argument is clamped at
// -96 (2^-96 is far below the resolution of every sum it enters) so the exponent arithmetic cannot underflow.
// Round-to-nearest split through the 1.5*2^52 constant (the integer lands in the low mantissa word), degree-10
// interpolant of 2^f on |f| <= 1/2, exponent added with integer arithmetic.
__device__ __forceinline__ double exp2_nonpositive(double y) {
  y = fmax(y, -96.0);
  const double kShift = 6755399441055744.0;
  const double t = y + kShift;
  const double f = y - (t - kShift);
  double p = 0x1.e6063f7217bc6p-28;
  p = fma(p, f, 0x1.b675bca4eeebbp-24);
  p = fma(p, f, 0x1.62bfd47773353p-20);
  p = fma(p, f, 0x1.ffcb54062e698p-17);
  p = fma(p, f, 0x1.430913096fd9fp-13);
  p = fma(p, f, 0x1.5d87fe9d7a584p-10);
  p = fma(p, f, 0x1.3b2ab6fba1ddap-7);
  p = fma(p, f, 0x1.c6b08d703ce49p-5);
  p = fma(p, f, 0x1.ebfbdff82c598p-3);
  p = fma(p, f, 0x1.62e42fefa3a19p-1);
  p = fma(p, f, 1.0);
  const int n = __double2loint(t);
  return __hiloint2double(__double2hiint(p) + (n << 20), __double2loint(p));
}

// 1/sqrt(u) for normal positive u: hardware seed and one third-order correction (error ~ seed error cubed)
__device__ __forceinline__ double rsqrt_pos(double u) {
  const double y0 = __builtin_amdgcn_rsq(u);
  const double e = fma(-(u * y0), y0, 1.0);
  return fma(y0 * e, fma(0.375, e, 0.5), y0);
}

// exp(x) for x <= 0 through the same core
__device__ __forceinline__ double exp_nonpositive(double x) { return exp2_nonpositive(x * 1.4426950408889634074); }

// ---- deterministic mode ---------------------------------------------------------------------------------------
// Run-to-run differences come from floating-point sums whose ORDER is not fixed: FP64 atomics (LDS and HBM), terms
// that reach a lane in an order decided by atomic cursors, and the composition of a forest (decided by the order in
// which the bookkeeping's atomics are served).  In deterministic mode every term that enters such a sum is first
// rounded to a multiple of 2^-k.  Sums of multiples of 2^-k are exact in FP64 as long as they stay below 2^(53-k),
// and exact sums do not depend on their order -- the same ds_add_f64 / global_atomic_add_f64 then give bit-identical
// results, and nothing downstream changes.  (Beyond the range a sum merely rounds again, as in the default mode.)
// Quantum per kind of sum:                                  exact while |sum| <
constexpr double kQGrad = 17179869184.0;        // 2^34  forces, gradients (kJ/mol/nm): 5.8e-11   5.2e5
constexpr double kQVol = 4503599627370496.0;    // 2^52  self volumes (nm^3): 2.2e-16             2
constexpr double kQBorn = 17592186044416.0;     // 2^44  descreening sums (1/nm): 5.7e-14         512
constexpr double kQSum = 1099511627776.0;       // 2^40  GB Y sums, W+U sums: 9.1e-13             8192
constexpr double kQEnergy = 68719476736.0;      // 2^36  energies (kJ/mol): 1.5e-11               1.3e5
__device__ __forceinline__ double quantize(double v, double scale, bool det) { return det ? rint(v * scale) * (1.0 / scale) : v; }

// q^(3/2) for normal positive q: q^2 / sqrt(q)
__device__ __forceinline__ double pow_three_halves(double q) { return (q * q) * rsqrt_pos(q); }

}  // namespace agbnp
