// projections.hip -- equirectangular -> rectilinear (inverse gnomonic) remap.
//
// Replaces Projections::GnomonicProjection (src/projections.cc:51-86) and
// gnomonic_kernel (src/projections_program.cl:7-47).  Float builtins are
// evaluated as correctly rounded floats (DESIGN.md "Float model"); the two
// gaze-only angles and their sin/cos are computed once on the host.
#include <cmath>

#include "cr_math.h"
#include "f360_internal.h"

namespace {

#define F360_PI 3.141592653589793
#define F360_PI_2 1.5707963267948966

__device__ __forceinline__ float cr_atanf(float x) { return (float)atan((double)x); }
__device__ __forceinline__ float cr_sinf(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cr_cosf(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float cr_asinf(float x) { return (float)asin((double)x); }
__device__ __forceinline__ float cr_atan2f(float y, float x) {
  return (float)atan2((double)y, (double)x);
}

// fmod(x, 2 pi), exactly, without the library's iterative remainder loop wherever x lies in
// [8 pi, 14 pi) -- which is where the kernel's two calls (phi + pi/2 + 10 pi and
// lambda + pi + 10 pi, projections_program.cl:36-37) land for any gaze inside the frame; a gaze
// far outside it takes the library routine.  fmod returns x - n y with n = trunc(x / y), which
// is representable; in the window n is 4, 5 or 6, and every step below is a Sterbenz
// subtraction (operands within a factor of two of each other, hence exact): x - 4y (4y is a power-of-two multiple of y), then - 2y or
// - y as long as the remainder allows.
__device__ __forceinline__ double fmod_two_pi_window(double x) {
  const double y = 2 * F360_PI;
  if (!(x >= 4.0 * y && x < 7.0 * y)) return fmod(x, y);
  double a = x - 4.0 * y;
  if (a >= 2.0 * y) a -= 2.0 * y;
  if (a >= y) a -= y;
  return a;
}

// What the kernel computes from the target pixel alone (:21-24,29-30 and the sin / cos of c in
// :31-34): screen coordinates, rho, sin(atan(rho)), cos(atan(rho)).  Three of the five
// transcendentals per pixel do not depend on the view centre, so they are tabulated once per
// target geometry ("gnomonic.table") with this very code and read back every frame.  (All five
// values are stored: with x, y and rho recomputed -- two correctly rounded divisions and a
// square root -- and only sc, cc read back, the kernel is slower, 59 -> 66 us for a 3840x1920
// viewport; it is bound by instructions, the double asin and atan2 above all, not by bytes.)
struct GnomonicPixel {
  float x, y, rho, sc, cc;
};
__device__ __forceinline__ GnomonicPixel gnomonic_pixel(int i, int j, int dst_w, int dst_h) {
  GnomonicPixel p;
  p.x = 6.0f * ((float)i / (float)dst_w - 0.5f);  // scale = (6, 3)
  p.y = 3.0f * ((float)j / (float)dst_h - 0.5f);
  p.rho = sqrtf(p.x * p.x + p.y * p.y);
  const float c = cr_atanf(p.rho);
  p.sc = cr_sinf(c);
  p.cc = cr_cosf(c);
  return p;
}

// TABLE 1: five planes (x, y, rho, sc, cc); TABLE 2: two (sc, cc) -- x, y and rho are then
// recomputed per frame (two correctly rounded divisions and a square root), which pays once the
// kernel is no longer bound by the double-precision library routines (FAST below).
template <int TABLE>
__global__ __launch_bounds__(256) void gnomonic_table_kernel(float *__restrict__ table, int dst_w,
                                                            int dst_h) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= dst_w || j >= dst_h) return;
  const GnomonicPixel p = gnomonic_pixel(i, j, dst_w, dst_h);
  // planar layout: planes of dst_w * dst_h floats, so that every read is coalesced
  const size_t n = (size_t)dst_w * dst_h, at = (size_t)j * dst_w + i;
  if (TABLE == 1) {
    table[at] = p.x;
    table[n + at] = p.y;
    table[2 * n + at] = p.rho;
    table[3 * n + at] = p.sc;
    table[4 * n + at] = p.cc;
  } else {
    table[at] = p.sc;
    table[n + at] = p.cc;
  }
}

// FAST: asin and atan2 through cr_math.h -- a cheap double evaluation whose float rounding is
// accepted only when it is certainly the correctly rounded one; the few lanes where it is not
// (about 3 in 100,000) take the library routine, as every lane did before.
template <int TABLE, bool FAST>
__global__ __launch_bounds__(256) void gnomonic_kernel(
    uint32_t *__restrict__ dst, int dst_w, int dst_h,
    const uint32_t *__restrict__ src, int src_w, int src_h, float lambda0,
    float sp1, float cp1, const float *__restrict__ table) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= dst_w || j >= dst_h) return;
  GnomonicPixel p;
  if (TABLE == 1) {
    const size_t n = (size_t)dst_w * dst_h, at = (size_t)j * dst_w + i;
    p.x = table[at];
    p.y = table[n + at];
    p.rho = table[2 * n + at];
    p.sc = table[3 * n + at];
    p.cc = table[4 * n + at];
  } else if (TABLE == 2) {
    const size_t n = (size_t)dst_w * dst_h, at = (size_t)j * dst_w + i;
    p.x = 6.0f * ((float)i / (float)dst_w - 0.5f);
    p.y = 3.0f * ((float)j / (float)dst_h - 0.5f);
    p.rho = sqrtf(p.x * p.x + p.y * p.y);
    p.sc = table[at];
    p.cc = table[n + at];
  } else {
    p = gnomonic_pixel(i, j, dst_w, dst_h);
  }
  const float x = p.x, y = p.y, rho = p.rho, sc = p.sc, cc = p.cc;
  const float asin_arg = cc * sp1 + (y * sc * cp1) / rho;
  const float at_y = x * sc, at_x = rho * cp1 * cc - y * sp1 * sc;
  float phi, at2;
  if (FAST) {
    bool ok_a, ok_t;
    phi = f360::cr_asinf_fast(asin_arg, ok_a);
    at2 = f360::cr_atan2f_fast(at_y, at_x, ok_t);
    if (!ok_a) phi = cr_asinf(asin_arg);
    if (!ok_t) at2 = cr_atan2f(at_y, at_x);
  } else {
    phi = cr_asinf(asin_arg);
    at2 = cr_atan2f(at_y, at_x);
  }
  float lam = lambda0 + at2;
  phi = (float)fmod_two_pi_window((double)phi + F360_PI_2 + 10 * F360_PI);
  lam = (float)fmod_two_pi_window((double)lam + F360_PI + 10 * F360_PI);
  float su = (float)((double)lam / (2.0 * F360_PI));
  float sv = (float)((double)phi / (F360_PI));
  // clamp() = fmin(fmax(x, lo), hi): the NaN of the exact viewport centre
  // (rho == 0 -> 0/0) clamps to 0
  su = fminf(fmaxf(su, 0.0f), 0.999f);
  sv = fminf(fmaxf(sv, 0.0f), 0.999f);
  const size_t texel =
      (size_t)(int)(sv * (float)src_h) * src_w + (int)(su * (float)src_w);
  dst[(size_t)j * dst_w + i] = src[texel] & 0x00ffffffu;
}

// Debug / test entry: the fast routines on arrays (kind 0: asin(a), 1: atan2(a, b)); out = the
// float they return, flag = 1 where they vouch for it.
__global__ __launch_bounds__(256) void cr_math_probe_kernel(int kind, size_t n,
                                                           const float *__restrict__ a,
                                                           const float *__restrict__ b,
                                                           float *__restrict__ out,
                                                           uint8_t *__restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool ok;
  out[i] = kind == 0 ? f360::cr_asinf_fast(a[i], ok) : f360::cr_atan2f_fast(a[i], b[i], ok);
  flag[i] = ok ? 1 : 0;
}

}  // namespace

extern "C" int f360_gnomonic(f360_ctx *ctx, uint8_t *target_dev, int target_width,
                             int target_height, int target_linesize,
                             const uint8_t *source_dev, int source_width,
                             int source_height, int source_linesize,
                             float center_x, float center_y) {
  (void)target_linesize;  // 4-byte texels, tightly packed rows, as in the kernel
  (void)source_linesize;
  F360_REQUIRE(ctx, "f360_gnomonic: null context");
  F360_BIND_DEVICE(ctx);
  F360_REQUIRE(target_dev && source_dev, "f360_gnomonic: null buffer");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 1 &&
                   source_height >= 1,
               "f360_gnomonic: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_gnomonic: a dimension exceeds 65536");
  F360_REQUIRE(((uintptr_t)target_dev % 4) == 0 && ((uintptr_t)source_dev % 4) == 0,
               "f360_gnomonic: buffers must be 4-byte aligned");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_gnomonic: gaze centre out of range");
  // float phi1 = (center.y - 0.5) * PI; float lambda0 = (center.x - 0.5) * 2.0 * PI;
  const float phi1 = (float)(((double)center_y - 0.5) * F360_PI);
  const float lambda0 = (float)(((double)center_x - 0.5) * 2.0 * F360_PI);
  const float sp1 = (float)std::sin((double)phi1);
  const float cp1 = (float)std::cos((double)phi1);
  const dim3 grid((target_width + 63) / 64, (target_height + 3) / 4);
  const float *table = nullptr;
  // "gnomonic.table": 0 none, 1 five planes (x, y, rho, sc, cc), 2 two planes (sc, cc)
  int table_kind = ctx->opt_gnomonic_table;
  const size_t table_bytes =
      (size_t)target_width * target_height * (table_kind == 2 ? 2 : 5) * sizeof(float);
  if (table_bytes > ((size_t)1 << 30)) table_kind = 0;
  if (table_kind) {
    if (ctx->gn_w != target_width || ctx->gn_h != target_height || ctx->gn_kind != table_kind ||
        !ctx->gn_table.p) {
      F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // earlier calls may read the old table
      int st = ctx->gn_table.reserve(table_bytes);
      if (st != F360_OK) return st;
      if (table_kind == 2)
        hipLaunchKernelGGL(gnomonic_table_kernel<2>, grid, dim3(256), 0, ctx->stream,
                           ctx->gn_table.as<float>(), target_width, target_height);
      else
        hipLaunchKernelGGL(gnomonic_table_kernel<1>, grid, dim3(256), 0, ctx->stream,
                           ctx->gn_table.as<float>(), target_width, target_height);
      F360_HIP_TRY(hipGetLastError());
      ctx->gn_w = target_width;
      ctx->gn_h = target_height;
      ctx->gn_kind = table_kind;
    }
    table = ctx->gn_table.as<float>();
  }
  f360::KernelSpan span(ctx, f360::kGnomonic, f360::take_profile_slot(ctx));
  uint32_t *dst = reinterpret_cast<uint32_t *>(target_dev);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(source_dev);
#define F360_GN_LAUNCH(T, F)                                                                     \
  hipLaunchKernelGGL((gnomonic_kernel<T, F>), grid, dim3(256), 0, ctx->stream, dst, target_width, \
                     target_height, src, source_width, source_height, lambda0, sp1, cp1, table)
  const bool fast = ctx->opt_gnomonic_fast != 0;
  if (table_kind == 1) {
    if (fast) F360_GN_LAUNCH(1, true); else F360_GN_LAUNCH(1, false);
  } else if (table_kind == 2) {
    if (fast) F360_GN_LAUNCH(2, true); else F360_GN_LAUNCH(2, false);
  } else {
    if (fast) F360_GN_LAUNCH(0, true); else F360_GN_LAUNCH(0, false);
  }
#undef F360_GN_LAUNCH
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

// Test entry for csrc/cr_math.h: kind 0 asin(a_dev[i]), kind 1 atan2(a_dev[i], b_dev[i]) through
// the fast routines; out_dev[i] = the float they return, flag_dev[i] = 1 where they vouch that it
// is the correctly rounded one.  Not part of the reference surface.
extern "C" int f360_debug_cr_math(f360_ctx *ctx, int kind, size_t n, const float *a_dev,
                                  const float *b_dev, float *out_dev, uint8_t *flag_dev) {
  F360_REQUIRE(ctx && a_dev && out_dev && flag_dev && (kind == 0 || (kind == 1 && b_dev)),
               "f360_debug_cr_math: bad argument");
  F360_BIND_DEVICE(ctx);
  if (n == 0) return F360_OK;
  hipLaunchKernelGGL(cr_math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, kind, n, a_dev, b_dev, out_dev, flag_dev);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
