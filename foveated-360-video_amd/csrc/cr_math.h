// cr_math.h -- fast correctly rounded float asin / atan2 for the non-separable remaps.
//
// The oracle (and therefore this engine) defines every OpenCL float builtin of the reference's
// kernels as the CORRECTLY ROUNDED float of the exact result: evaluate in double, round once
// (DESIGN.md "Float model").  The library's double asin / atan2 that the kernels used for this
// cost ~450 double-precision instructions per pixel and left the gnomonic remap bound by
// instructions at 0.13 of the roofline.  What is needed is much less than a double result: a
// FLOAT, i.e. 24 bits, right in all but vanishingly few cases -- and to know when it is not.
//
//   1. r~ = a cheap double approximation of the exact result with a proven RELATIVE error
//      bound E (below);
//   2. the correctly rounded float is (float)r~ whenever rounding r~ (1 - EPS) and r~ (1 + EPS)
//      gives the same float (rounding is monotonic and the exact result lies between them);
//   3. otherwise -- r~ within EPS of a rounding boundary: about 3 in 100,000 arguments -- the
//      caller takes the library routine (`ok` is false).
//
// One core serves both functions: atan2(y, x) for finite, not both zero arguments, and
// asin(a) = atan2(a, sqrt((1 - a)(1 + a))) (both factors exact in double for a float a).
//
//   atan2 core: t = min(|x|,|y|) / max(|x|,|y|) in [0, 1]; with u = t for t <= tan(pi/8) and
//   u = (t - 1)/(t + 1) (= (mn - mx)/(mn + mx): still ONE division) beyond, |u| <= tan(pi/8) and
//   atan(t) = [pi/4 +] u Q(u^2), Q of degree 9 (tools/fit_atan.py: Chebyshev interpolant
//   computed in 80-bit arithmetic); octant and sign fix-ups with pi/2, pi.
//
// Error budget (relative to the result, which may be tiny: atan2(y, x) ~ y/x): Q interpolates
// atan(u)/u itself, so u Q(u^2) is within 2e-15 of atan(u) RELATIVELY on the whole range
// (tools/fit_atan.py evaluates the kernel's own Horner order on 8 million points against long
// double atan; tests/test_cr_math.py repeats it); the division and the subtraction / addition in
// front of it <= 4 ulp of t: 5e-16; sqrt for asin 2e-16; the pi/4, pi/2, pi constants and their
// additions 1e-15 absolute on results >= pi/8: E < 5e-15.  EPS = 1e-12 leaves a factor of 200.
// tests/test_gpu_cr_math.py runs 2^24 random and all the boundary arguments through the device
// functions against the oracle's definition, bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace f360 {

constexpr double kCrEps = 1e-12;
constexpr double kTanPi8 = 0.41421356237309503;

// tools/fit_atan.py --terms 10 --emit
__device__ __forceinline__ double cr_atan_poly(double u) {
  const double z = u * u;
  double q = -0x1.9eba5a9e61f6bp-6;
  q = fma(q, z, 0x1.9b9b8b40b97a7p-5);
  q = fma(q, z, -0x1.0a7b5d0274882p-4);
  q = fma(q, z, 0x1.3a4eb7e4e28c6p-4);
  q = fma(q, z, -0x1.744e9df9edc50p-4);
  q = fma(q, z, 0x1.c71bca7ab0dc2p-4);
  q = fma(q, z, -0x1.249246f5db1b8p-3);
  q = fma(q, z, 0x1.999999922a3a3p-3);
  q = fma(q, z, -0x1.5555555550676p-2);
  q = fma(q, z, 0x1.ffffffffffff7p-1);
  return u * q;
}

// atan2(y, x) for doubles that came from finite floats, not both zero; relative error < 5e-15
__device__ __forceinline__ double cr_atan2_core(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double mn = fmin(ax, ay), mx = fmax(ax, ay);
  const bool far = mn > kTanPi8 * mx;  // t beyond tan(pi/8): rotate by pi/4
  const double num = far ? mn - mx : mn;
  const double den = far ? mn + mx : mx;
  double r = cr_atan_poly(num / den);
  if (far) r += 0.78539816339744831;            // pi/4
  if (ay > ax) r = 1.5707963267948966 - r;      // pi/2
  if (x < 0.0) r = 3.141592653589793 - r;       // pi  (x = -0 cannot reach here with r != 0
                                                //      ... see cr_atan2f_fast: rejected)
  return __builtin_signbit(y) ? -r : r;
}

// (float)r, and whether that is certainly the correctly rounded float of the exact result
__device__ __forceinline__ float cr_round_checked(double r, bool &ok) {
  const float f = (float)r;
  const double e = kCrEps * fabs(r);  // (r = +-0 is exact: atan2(+-0, x > 0), asin(+-0))
  ok = ok && (float)(r - e) == f && (float)(r + e) == f;
  return f;
}

// Correctly rounded atan2f(y, x) when `ok` comes back true.  Rejected up front (library path):
// non-finite arguments, both zero, and a zero x with the sign bit set (atan2 distinguishes -0
// from +0 there; not worth a fast path).
__device__ __forceinline__ float cr_atan2f_fast(float y, float x, bool &ok) {
  ok = isfinite(y) && isfinite(x) && (x != 0.0f || y != 0.0f) &&
       !(x == 0.0f && __builtin_signbit(x));
  return cr_round_checked(cr_atan2_core((double)y, (double)x), ok);
}

// Correctly rounded asinf(a) when `ok` comes back true; |a| > 1 and NaN are rejected.
__device__ __forceinline__ float cr_asinf_fast(float a, bool &ok) {
  ok = fabsf(a) <= 1.0f;
  const double d = (double)a;
  const double c = sqrt((1.0 - d) * (1.0 + d));  // both factors exact: a is a float
  // a = +-1: c = 0, the core sees (y = +-1, x = +0): t = 0, "ay > ax" -> pi/2
  return cr_round_checked(cr_atan2_core(d, c), ok);
}

}  // namespace f360
