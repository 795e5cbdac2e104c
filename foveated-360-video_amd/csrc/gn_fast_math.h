// gn_fast_math.h -- float asin / atan2 with known ABSOLUTE error bounds, for the index-guarded
// gnomonic remap (projections.hip, "gnomonic.guard").
//
// The remap ends in two integer texel indices, (int)(sv * source_height) and
// (int)(su * source_width) (src/projections_program.cl:40-43).  The kernel text reaches them
// through correctly rounded float asin / atan2 (the oracle's definition of the OpenCL builtins)
// and a few more float roundings; every one of those moves the final product by a bounded
// amount.  So the index is known for certain from a CHEAP evaluation of the same real-valued
// chain whenever that evaluation lands further from an integer than the sum of the bounds --
// and only the pixels that land closer (1.5 in 100 at 8K) need the exact chain.  The arguments of
// asin and atan2 are computed with the kernel text's own float operations, so they are the same
// floats in both evaluations: no condition number enters.
//
// What this file provides: the two transcendental cores in float (hardware rcp / sqrt, fused
// multiply-adds -- none of which the exact chain may use) and the bounds kGnEAsin / kGnEAtan2 on
// |result - true value|.  tools/fit_gn_fast.py derives the coefficients (Chebyshev interpolants
// in 80-bit arithmetic) and estimates the error on the host; tests/test_gpu_gn_fast.py sweeps the
// DEVICE functions -- every float in [-1, 1] for asin, 2^30 pseudo-random argument pairs over 60
// binades and all sign / octant combinations for atan2 -- against double precision and requires
// the largest error seen to stay below HALF the bound used.
#pragma once
#include <hip/hip_runtime.h>

namespace f360 {

constexpr float kGnEAsin = 4.0e-7f;   // |gn_asin_fast(a) - asin a|, |a| <= 1
constexpr float kGnEAtan2 = 8.0e-7f;  // |gn_atan2_fast(y, x) - atan2(y, x)| where it vouches

// tools/fit_gn_fast.py --atan-terms 9 --asin-terms 5 --emit
__device__ __forceinline__ float gn_atan_q(float z) {
  float q = 0x1.6a689ep-9f;
  q = __builtin_fmaf(q, z, -0x1.01a896p-6f);
  q = __builtin_fmaf(q, z, 0x1.5920d8p-5f);
  q = __builtin_fmaf(q, z, -0x1.31685ap-4f);
  q = __builtin_fmaf(q, z, 0x1.b2eadap-4f);
  q = __builtin_fmaf(q, z, -0x1.22c504p-3f);
  q = __builtin_fmaf(q, z, 0x1.996ef2p-3f);
  q = __builtin_fmaf(q, z, -0x1.55548ep-2f);
  q = __builtin_fmaf(q, z, 1.0f);
  return q;
}
__device__ __forceinline__ float gn_asin_r(float z) {
  float r = 0x1.382394p-5f;
  r = __builtin_fmaf(r, z, 0x1.b2f144p-6f);
  r = __builtin_fmaf(r, z, 0x1.70a8e4p-5f);
  r = __builtin_fmaf(r, z, 0x1.332726p-4f);
  r = __builtin_fmaf(r, z, 0x1.55555ep-3f);
  return r;
}

// pi/2 and pi as float pairs (hi + lo to ~2^-49)
constexpr float kPio2Hi = 0x1.921fb6p+0f, kPio2Lo = -0x1.777a5cp-25f;
constexpr float kPiHi = 0x1.921fb6p+1f, kPiLo = -0x1.777a5cp-24f;

// asin(a) for |a| <= 1 (anything else: the caller does not use the result)
__device__ __forceinline__ float gn_asin_fast(float a) {
  const float s = __builtin_fabsf(a);
  const bool small = s <= 0.5f;
  const float z = small ? a * a : (1.0f - s) * 0.5f;
  const float b = small ? s : __builtin_amdgcn_sqrtf(z);
  const float p = __builtin_fmaf(b * z, gn_asin_r(z), b);  // asin b (small) or asin sqrt(z)
  const float v = small ? p : (kPio2Hi - 2.0f * p) + kPio2Lo;
  return __builtin_copysignf(v, a);
}

// atan2(y, x); `ok` = false where the routine does not vouch for its bound (both zero,
// non-finite or extreme magnitudes: the caller takes the exact chain there)
__device__ __forceinline__ float gn_atan2_fast(float y, float x, bool &ok) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
  ok = mx >= 1.0e-30f && mx <= 1.0e30f;  // (NaN compares false)
  float rc = __builtin_amdgcn_rcpf(mx);
  rc = __builtin_fmaf(__builtin_fmaf(-mx, rc, 1.0f), rc, rc);  // one Newton step: ~0.5 ulp
  const float t = mn * rc;
  float r = t * gn_atan_q(t * t);
  if (ay > ax) r = (kPio2Hi - r) + kPio2Lo;
  if (x < 0.0f) r = (kPiHi - r) + kPiLo;
  return __builtin_copysignf(r, y);
}

}  // namespace f360
