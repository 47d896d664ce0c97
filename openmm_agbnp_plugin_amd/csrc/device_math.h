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

// q^(3/2) for normal positive q: q^2 / sqrt(q)
__device__ __forceinline__ double pow_three_halves(double q) { return (q * q) * rsqrt_pos(q); }

}  // namespace agbnp
