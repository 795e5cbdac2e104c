// projections.hip -- equirectangular -> rectilinear (inverse gnomonic) remap.
//
// Replaces Projections::GnomonicProjection (src/projections.cc:51-86) and
// gnomonic_kernel (src/projections_program.cl:7-47).  Float builtins are
// evaluated as correctly rounded floats (DESIGN.md "Float model"); the two
// gaze-only angles and their sin/cos are computed once on the host.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "gn_fast_math.h"
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

// Five planes (x, y, rho, sc, cc) of the exact kernel's view-independent terms.
__global__ __launch_bounds__(256) void gnomonic_table_kernel(float *__restrict__ table, int dst_w,
                                                            int dst_h) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= dst_w || j >= dst_h) return;
  const GnomonicPixel p = gnomonic_pixel(i, j, dst_w, dst_h);
  // planar layout: planes of dst_w * dst_h floats, so that every read is coalesced
  const size_t n = (size_t)dst_w * dst_h, at = (size_t)j * dst_w + i;
  table[at] = p.x;
  table[n + at] = p.y;
  table[2 * n + at] = p.rho;
  table[3 * n + at] = p.sc;
  table[4 * n + at] = p.cc;
}

// The kernel text from the asin on (:31-43) for one pixel, every float builtin correctly rounded:
// the texel index.
__device__ __forceinline__ size_t gnomonic_texel_exact(const GnomonicPixel &p, float lambda0,
                                                       float sp1, float cp1, int src_w,
                                                       int src_h) {
  const float x = p.x, y = p.y, rho = p.rho, sc = p.sc, cc = p.cc;
  const float asin_arg = cc * sp1 + (y * sc * cp1) / rho;
  const float at_y = x * sc, at_x = rho * cp1 * cc - y * sp1 * sc;
  float phi = cr_asinf(asin_arg);
  const float at2 = cr_atan2f(at_y, at_x);
  float lam = lambda0 + at2;
  phi = (float)fmod_two_pi_window((double)phi + F360_PI_2 + 10 * F360_PI);
  lam = (float)fmod_two_pi_window((double)lam + F360_PI + 10 * F360_PI);
  float su = (float)((double)lam / (2.0 * F360_PI));
  float sv = (float)((double)phi / (F360_PI));
  // clamp() = fmin(fmax(x, lo), hi): the NaN of the exact viewport centre
  // (rho == 0 -> 0/0) clamps to 0
  su = fminf(fmaxf(su, 0.0f), 0.999f);
  sv = fminf(fmaxf(sv, 0.0f), 0.999f);
  return (size_t)(int)(sv * (float)src_h) * src_w + (int)(su * (float)src_w);
}

// The exact chain on every pixel (any gaze, any size; "gnomonic.guard" = 0 or a viewport the
// guarded remap does not take).  TABLE: the view-independent terms come from the five planes.
template <int TABLE>
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
  } else {
    p = gnomonic_pixel(i, j, dst_w, dst_h);
  }
  const size_t texel = gnomonic_texel_exact(p, lambda0, sp1, cp1, src_w, src_h);
  dst[(size_t)j * dst_w + i] = src[texel] & 0x00ffffffu;
}

// ---------------------------------------------------------------------------------------------
// "gnomonic.guard" (the default): the index-guarded remap.  gn_fast_math.h says why it is exact.
//
// Error budget, as fractions of the source size (what the (int) casts of :41-42 see is that times
// source_height / source_width).  A = asin argument, Ty / Tx = atan2 arguments: computed with the
// kernel text's own float operations from the same x, y, rho, sc, cc -- identical floats in the
// exact chain and here.  RN = one float rounding, at most half an ulp of the result.
//   exact chain, v: phi0 = RN(asin A): 2^-24 rad (|phi0| < 2); the double additions and the fmod
//      are exact to 1e-14; phi = RN(.) of a value below 4: 2^-23 rad; sv = RN(phi / pi) <= 1:
//      2^-24; the product RN(sv * H): 2^-24.        (2^-24 + 2^-23) / pi + 2^-23 = 1.77e-7
//   exact chain, u: at2 = RN(atan2): 2^-23 rad (|.| < 4); lam0 = RN(lambda0 + at2) against
//      RN(lambda0 + fast): inputs within 2^-23 + kGnEAtan2, one more ulp of a value below 8
//      (2^-21) for the two roundings; lam = RN(.) of a value below 8: 2^-22; su: 2^-24; product
//      2^-24.                  (2^-23 + 2^-21 + 2^-22 + kGnEAtan2) / (2 pi) + 2^-23 = 3.8e-7
//      (|lambda0| <= pi, i.e. the gaze inside the frame; otherwise the launch is the exact one)
//   this side, v: kGnEAsin / pi; phi + pi/2 in float (result below 4: 2^-23 rad, the constant
//      4.4e-8) -> 5.2e-8; times RN(H / pi): two relative 2^-24.                     total 3.0e-7
//   this side, u: fma(lam0, RN(1 / 2 pi), 0.5): 2^-24 for the constant, 2^-24 for the result
//      (below 2); the fraction is exact; times W: 2^-24.                             total 1.8e-7
// The guards are 1.25 x the sums: 5.9e-7 H and 7.0e-7 W texels.  A pixel is ACCEPTED when both
// fast products lie further than the guard from every integer (0 and the size included: that
// also keeps the fast and the exact value on the same side of the fmod wrap) and further than
// 2e-6 of the size from the clamp at 0.999; beyond the clamp the index is the constant
// (int)(0.999f * size) -- for u only up to one guard before the wrap at the right edge (found
// by scripts/gn_guard_soak.py: the centre row of a view onto a pole runs along the seam).  Everything else -- about 1.5 pixels in 100, plus NaN / out-of-domain
// arguments -- goes through the exact chain, 64 at a time (see the kernel).  Built and measured
// on the way: a second launch over a global worklist (its exact pass is latency-bound at 15-17 us
// whatever the list length; one global counter serialised ~60,000 atomics: 560 us; 1024 lists:
// 37 + 17 us); the rejects of a 1024-thread workgroup compacted in LDS and worked off by its
// first lanes behind a barrier (53 us: the lone wave on the exact chain keeps the workgroup's
// slot for microseconds).
struct GnGuard {
  float kv, ku;          // RN(source_height / pi), source_width
  float dv, du;          // guards, in texels
  float clamp_lo_v, clamp_hi_v, clamp_lo_u, clamp_hi_u;  // (0.999 -+ 2e-6) * size
  int kclamp_v, kclamp_u;  // (int)(0.999f * size)
  uint32_t *debug_total;   // rejects of the launch (tests / scripts), or null
};

constexpr int kGnThreads = 256;

// planes of dst_w * dst_h floats: rho, sin(atan rho), cos(atan rho); then x[dst_w], y[dst_h]
__global__ __launch_bounds__(256) void gnomonic_guard_table_kernel(float *__restrict__ planes,
                                                                  float *__restrict__ xtab,
                                                                  float *__restrict__ ytab,
                                                                  int dst_w, int dst_h) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= dst_w || j >= dst_h) return;
  const GnomonicPixel p = gnomonic_pixel(i, j, dst_w, dst_h);
  const size_t n = (size_t)dst_w * dst_h, at = (size_t)j * dst_w + i;
  planes[at] = p.rho;
  planes[n + at] = p.sc;
  planes[2 * n + at] = p.cc;
  if (j == 0) xtab[i] = p.x;
  if (i == 0) ytab[j] = p.y;
}

// Persistent waves: a wave takes 64-pixel row segments in a strided loop and keeps the pixels it
// rejected in a wave-private LDS list; whenever 64 have come together it runs the exact chain on
// them with every lane busy, and once more at its end for the remainder.  No barrier, no atomic,
// and the exact chain's long latency (a wave alone on it takes microseconds) hides behind the
// other waves of the SIMD like any other.
__global__ __launch_bounds__(kGnThreads) void gnomonic_guard_kernel(
    uint32_t *__restrict__ dst, int dst_w, int dst_h, const uint32_t *__restrict__ src, int src_w,
    int src_h, float lambda0, float sp1, float cp1, const float *__restrict__ planes,
    const float *__restrict__ xtab, const float *__restrict__ ytab, const GnGuard g, int tiles_x,
    int ntiles) {
  __shared__ uint32_t lists[kGnThreads / 64][128];
  // (the wave index as a scalar: the tile arithmetic below -- a division -- then runs on the
  // scalar unit instead of in every lane)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *list = lists[wave];
  const uint32_t n = (uint32_t)dst_w * (uint32_t)dst_h;
  uint32_t pending = 0, total = 0;  // wave-uniform

  auto resolve = [&](uint32_t at) {
    const uint32_t pj = at / (uint32_t)dst_w, pi = at - pj * (uint32_t)dst_w;
    GnomonicPixel p;
    p.x = xtab[pi];
    p.y = ytab[pj];
    p.rho = planes[at];
    p.sc = planes[n + at];
    p.cc = planes[2 * n + at];
    const size_t texel = gnomonic_texel_exact(p, lambda0, sp1, cp1, src_w, src_h);
    dst[at] = src[texel] & 0x00ffffffu;
  };

  // Workgroups go round the 8 XCDs, each with its own L2.  With tiles dealt out in launch order
  // every XCD walks the whole viewport and pulls the whole source region through its L2 (PMC:
  // 265 MB fetched for 118 MB of table and texels); instead XCD k takes the k-th band of rows.
  const int nx = (gridDim.x & 7) == 0 ? 8 : 1;
  const int xcd = blockIdx.x % nx, local_wg = blockIdx.x / nx;
  const int row0 = (int)((long long)dst_h * xcd / nx), row1 = (int)((long long)dst_h * (xcd + 1) / nx);
  const int band_tiles = (row1 - row0) * tiles_x;
  const int nwaves = (gridDim.x / nx) * (kGnThreads / 64);
  (void)ntiles;
  // Two memory round trips per tile (table values, then the texel) and four waves per SIMD (the
  // exact chain's registers) leave the SIMDs waiting two thirds of the time if a wave runs them
  // back to back.  So the loop is a pipeline: the table values of the NEXT tile are requested
  // before the current one is computed, and the texels of the current one are stored an
  // iteration later.  Every load is unconditional (clamped addresses).
  struct TileIn {
    float x, y, rho, sc, cc;
    uint32_t at;
    bool valid;
  };
  auto request = [&](int tile, TileIn &t) {
    const int tc = min(tile, band_tiles - 1);
    const int jr = tc / tiles_x;
    const int j = row0 + jr, i = (tc - jr * tiles_x) * 64 + lane;
    const int ic = min(i, dst_w - 1);
    t.valid = tile < band_tiles && i < dst_w;
    t.at = (uint32_t)j * (uint32_t)dst_w + (uint32_t)ic;
    t.x = xtab[ic];
    t.y = ytab[j];
    t.rho = planes[t.at];
    t.sc = planes[n + t.at];
    t.cc = planes[2 * n + t.at];
  };
  uint32_t held_texel = 0, held_at = 0;  // the previous tile's texel, on its way
  bool held = false;
  TileIn cur;
  int tile = local_wg * (kGnThreads / 64) + wave;
  if (band_tiles > 0) request(tile, cur);
  for (; tile < band_tiles; tile += nwaves) {
    TileIn next;
    request(tile + nwaves, next);
    bool ok = true;  // (lanes beyond the row end have nothing to do)
    uint32_t texel_index = 0;
    const uint32_t at = cur.at;
    {
      const float x = cur.x, y = cur.y, rho = cur.rho, sc = cur.sc, cc = cur.cc;
      // the kernel text's float operations (:31-34), the same floats the exact chain sees
      const float asin_arg = cc * sp1 + (y * sc * cp1) / rho;
      const float at_y = x * sc, at_x = rho * cp1 * cc - y * sp1 * sc;
      const float at_f = f360::gn_atan2_fast(at_y, at_x, ok);
      ok = ok && __builtin_fabsf(asin_arg) <= 1.0f;  // (NaN at the exact viewport centre: rejected)
      const float phi_f = f360::gn_asin_fast(asin_arg);
      const float lam0 = lambda0 + at_f;
      const float fv = (phi_f + f360::kPio2Hi) * g.kv;
      const float ru = __builtin_fmaf(lam0, 0x1.45f306p-3f, 0.5f);  // lam0 / 2 pi + 11/2
      const float fu = (ru - __builtin_floorf(ru)) * g.ku;
      const float flv = __builtin_floorf(fv), flu = __builtin_floorf(fu);
      const float frv = fv - flv, fru = fu - flu;
      const bool in_v = frv >= g.dv && frv <= 1.0f - g.dv && fv < g.clamp_lo_v;
      const bool in_u = fru >= g.du && fru <= 1.0f - g.du && fu < g.clamp_lo_u;
      // (beyond the clamp u still has the fmod wrap at the right edge ahead of it: a value that
      // the exact chain may already have wrapped to 0 must not be taken for a clamped one)
      const bool top_v = fv > g.clamp_hi_v, top_u = fu > g.clamp_hi_u && fu < g.ku - g.du;
      const int iy = top_v ? g.kclamp_v : (int)flv;
      const int ix = top_u ? g.kclamp_u : (int)flu;
      ok = ok && (in_v || top_v) && (in_u || top_u);
      // (a rejected lane's indices may be anything: clamp them, its texel is not used)
      texel_index = __umul24((uint32_t)min(max(iy, 0), src_h - 1), (uint32_t)src_w) +
                    (uint32_t)min(max(ix, 0), src_w - 1);  // (24-bit multiply: full rate)
    }
    if (held) dst[held_at] = held_texel & 0x00ffffffu;  // the previous tile's pixels
    held_texel = src[texel_index];
    held_at = at;
    held = ok && cur.valid;
    const unsigned long long rejected = __ballot(!ok && cur.valid);
    if (rejected) {
      if (!ok && cur.valid)
        list[pending + (uint32_t)__popcll(rejected & ((1ull << lane) - 1))] = at;
      pending += (uint32_t)__popcll(rejected);
      total += (uint32_t)__popcll(rejected);
      if (pending >= 64) {
        // the list is wave-private and a wave's LDS operations execute in order; the fence only
        // keeps the compiler from moving the reads of other lanes' entries above the writes
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        pending -= 64;
        resolve(list[pending + lane]);
      }
    }
    cur = next;
  }
  if (held) dst[held_at] = held_texel & 0x00ffffffu;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < (int)pending) resolve(list[lane]);
  if (g.debug_total && lane == 0 && total) atomicAdd(g.debug_total, total);
}

// Test entry: sweeps of the fast cores against double precision; the largest absolute error
// seen, as float bits through atomicMax (errors are non-negative).
//   kind 0: gn_asin_fast over EVERY float in [-1, 1]
//   kind 1: gn_atan2_fast over `n` pseudo-random pairs (exponents -30 .. 30, all signs) and,
//           for the first 2^26 indices, ratios swept densely through [0, 1] in every octant
__device__ __forceinline__ uint32_t gn_hash(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__global__ __launch_bounds__(256) void gn_fast_sweep_kernel(int kind, unsigned long long n,
                                                           uint32_t *__restrict__ worst) {
  float err_max = 0.0f;
  for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < n;
       k += (unsigned long long)gridDim.x * 256) {
    if (kind == 0) {
      // bit patterns 0 .. 0x3f800000 are the floats 0 .. 1; the upper half of k is the sign
      const uint32_t pat = (uint32_t)(k % 0x3f800001ull);
      float a = __uint_as_float(pat);
      if (k >= 0x3f800001ull) a = -a;
      const double e = fabs((double)f360::gn_asin_fast(a) - asin((double)a));
      err_max = fmaxf(err_max, (float)e);
    } else {
      const uint32_t h1 = gn_hash((uint32_t)k * 2u + 1u), h2 = gn_hash((uint32_t)(k >> 32) + h1);
      float x, y;
      if (k < (1ull << 26)) {
        // dense ratios: t = k / 2^23 stepped over [0, 1] at 8 magnitudes, octant from h2
        const float t = (float)(k & 0x7fffffu) * (1.0f / 8388608.0f);
        const float m = __uint_as_float(((uint32_t)(97 + 8 * (int)(k >> 23))) << 23);
        x = m;
        y = m * t;
        if (h2 & 1u) { const float q = x; x = y; y = q; }
      } else {
        x = __uint_as_float((((h1 >> 8) % 61u + 97u) << 23) | (h2 & 0x7fffffu));
        y = __uint_as_float((((h1 >> 16) % 61u + 97u) << 23) | (gn_hash(h2) & 0x7fffffu));
      }
      if (h2 & 2u) x = -x;
      if (h2 & 4u) y = -y;
      bool ok;
      const float r = f360::gn_atan2_fast(y, x, ok);
      if (ok) {
        const double e = fabs((double)r - atan2((double)y, (double)x));
        err_max = fmaxf(err_max, (float)e);
      }
    }
  }
  atomicMax(worst, __float_as_uint(err_max));
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
  uint32_t *dst = reinterpret_cast<uint32_t *>(target_dev);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(source_dev);
  // "gnomonic.guard": the index-guarded remap, for a gaze inside the frame
  // (|lambda0| <= pi is what its error budget assumes) and a table of at most 1 GiB (12 bytes
  // per viewport pixel; its 32-bit plane indices 2 * npix + at stay far below 2^32 then).  A
  // larger viewport, or one whose table cannot be allocated, takes the exact chain below.
  const size_t npix = (size_t)target_width * target_height;
  const size_t gtab_bytes = (3 * npix + target_width + target_height) * sizeof(float);
  bool guarded = ctx->opt_gnomonic_guard && std::fabs(lambda0) <= 3.1415928f &&
                 gtab_bytes <= ((size_t)1 << 30);
  if (guarded && (ctx->gn_gw != target_width || ctx->gn_gh != target_height || !ctx->gn_gtab.p)) {
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // earlier calls may read the old tables
    ctx->gn_gw = ctx->gn_gh = 0;
    if (ctx->gn_gtab.reserve(gtab_bytes) != F360_OK ||
        (!ctx->gn_counters.p && ctx->gn_counters.reserve(64) != F360_OK)) {
      (void)hipGetLastError();  // out of memory is not sticky; the exact chain needs no table
      guarded = false;
    }
  }
  if (guarded) {
    if (ctx->gn_gw != target_width || ctx->gn_gh != target_height) {
      float *t = ctx->gn_gtab.as<float>();
      hipLaunchKernelGGL(gnomonic_guard_table_kernel, grid, dim3(256), 0, ctx->stream, t,
                         t + 3 * npix, t + 3 * npix + target_width, target_width, target_height);
      F360_HIP_TRY(hipGetLastError());
      ctx->gn_gw = target_width;
      ctx->gn_gh = target_height;
    }
    const double e24 = 0x1p-24, e23 = 0x1p-23, e22 = 0x1p-22, e21 = 0x1p-21;
    const double exact_v = (e24 + e23) / F360_PI + e23;
    const double exact_u = (e23 + e21 + e22 + (double)f360::kGnEAtan2) / (2.0 * F360_PI) + e23;
    const double fast_v = (double)f360::kGnEAsin / F360_PI + (e23 + 4.4e-8) / F360_PI + 2 * e24;
    const double fast_u = 3 * e24;
    GnGuard g;
    g.kv = (float)((double)source_height / F360_PI);
    g.ku = (float)source_width;
    g.dv = (float)(1.25 * source_height * (exact_v + fast_v));
    g.du = (float)(1.25 * source_width * (exact_u + fast_u));
    g.clamp_lo_v = (float)((0.999 - 2e-6) * source_height);
    g.clamp_hi_v = (float)((0.999 + 2e-6) * source_height);
    g.clamp_lo_u = (float)((0.999 - 2e-6) * source_width);
    g.clamp_hi_u = (float)((0.999 + 2e-6) * source_width);
    g.kclamp_v = (int)(0.999f * (float)source_height);
    g.kclamp_u = (int)(0.999f * (float)source_width);
    // "debug.ablate" bit 9: count the rejected pixels of the launch (f360_debug_gnomonic_worklist)
    g.debug_total = nullptr;
    if (ctx->opt_ablate & 512) {
      g.debug_total = ctx->gn_counters.as<uint32_t>();
      F360_HIP_TRY(hipMemsetAsync(g.debug_total, 0, 4, ctx->stream));
    }
    const float *t = ctx->gn_gtab.as<float>();
    const int tiles_x = (target_width + 63) / 64, ntiles = tiles_x * target_height;
    if (!ctx->gn_cus) {
      int dev = 0, cus = 0;
      F360_HIP_TRY(hipGetDevice(&dev));
      F360_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      ctx->gn_cus = cus > 0 ? cus : 256;
      int a = 0;
      F360_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &a, reinterpret_cast<const void *>(gnomonic_guard_kernel), kGnThreads, 0));
      ctx->gn_wg_per_cu = std::max(a, 1);
    }
    // exactly as many workgroups as are resident at once (the exact chain's registers decide)
    // (a multiple of 8 where the viewport is large enough: one band of rows per XCD)
    int nwg = std::min((ntiles + kGnThreads / 64 - 1) / (kGnThreads / 64),
                       ctx->gn_cus * ctx->gn_wg_per_cu);
    if (nwg >= 64) nwg &= ~7;
    const dim3 ggrid(nwg);
    f360::KernelSpan span(ctx, f360::kGnomonic, f360::take_profile_slot(ctx));
    hipLaunchKernelGGL(gnomonic_guard_kernel, ggrid, dim3(kGnThreads), 0, ctx->stream, dst,
                       target_width, target_height, src, source_width, source_height, lambda0,
                       sp1, cp1, t, t + 3 * npix, t + 3 * npix + target_width, g, tiles_x, ntiles);
    F360_HIP_TRY(hipGetLastError());
    return F360_OK;
  }
  // the exact chain on every pixel; "gnomonic.table": its view-independent terms from five
  // planes (20 bytes per viewport pixel, at most 1 GiB; without it they are recomputed)
  const float *table = nullptr;
  const size_t table_bytes = npix * 5 * sizeof(float);
  if (ctx->opt_gnomonic_table && table_bytes <= ((size_t)1 << 30)) {
    if (ctx->gn_w != target_width || ctx->gn_h != target_height || !ctx->gn_table.p) {
      F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // earlier calls may read the old table
      ctx->gn_w = ctx->gn_h = 0;
      if (ctx->gn_table.reserve(table_bytes) == F360_OK) {
        hipLaunchKernelGGL(gnomonic_table_kernel, grid, dim3(256), 0, ctx->stream,
                           ctx->gn_table.as<float>(), target_width, target_height);
        F360_HIP_TRY(hipGetLastError());
        ctx->gn_w = target_width;
        ctx->gn_h = target_height;
      } else {
        (void)hipGetLastError();
      }
    }
    if (ctx->gn_w == target_width && ctx->gn_h == target_height) table = ctx->gn_table.as<float>();
  }
  f360::KernelSpan span(ctx, f360::kGnomonic, f360::take_profile_slot(ctx));
  if (table)
    hipLaunchKernelGGL(gnomonic_kernel<1>, grid, dim3(256), 0, ctx->stream, dst, target_width,
                       target_height, src, source_width, source_height, lambda0, sp1, cp1, table);
  else
    hipLaunchKernelGGL(gnomonic_kernel<0>, grid, dim3(256), 0, ctx->stream, dst, target_width,
                       target_height, src, source_width, source_height, lambda0, sp1, cp1, table);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

// Test entries for csrc/gn_fast_math.h and the guarded remap: the largest absolute error of a fast core
// over the sweep of `kind` (see gn_fast_sweep_kernel; blocks until done), the bounds the guard
// uses, and how many pixels of the last counted launch took the exact chain.  Not part of the
// reference surface.
extern "C" int f360_debug_gn_fast_sweep(f360_ctx *ctx, int kind, unsigned long long n,
                                        float *worst_out, float *bound_out) {
  F360_REQUIRE(ctx && worst_out && bound_out && (kind == 0 || kind == 1),
               "f360_debug_gn_fast_sweep: bad argument");
  F360_BIND_DEVICE(ctx);
  if (kind == 0) n = 2ull * 0x3f800001ull;
  uint32_t *d = nullptr;
  F360_HIP_TRY(hipMalloc(&d, 4));
  F360_HIP_TRY(hipMemsetAsync(d, 0, 4, ctx->stream));
  hipLaunchKernelGGL(gn_fast_sweep_kernel, dim3(4096), dim3(256), 0, ctx->stream, kind, n, d);
  uint32_t bits = 0;
  hipError_t e = hipMemcpyAsync(&bits, d, 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  F360_HIP_TRY(e);
  std::memcpy(worst_out, &bits, 4);
  *bound_out = kind == 0 ? f360::kGnEAsin : f360::kGnEAtan2;
  return F360_OK;
}

extern "C" int f360_debug_gnomonic_worklist(f360_ctx *ctx, unsigned *count_out) {
  F360_REQUIRE(ctx && count_out, "f360_debug_gnomonic_worklist: bad argument");
  F360_BIND_DEVICE(ctx);
  *count_out = 0;
  if (!ctx->gn_counters.p) return F360_OK;
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t c = 0;
  F360_HIP_TRY(hipMemcpy(&c, ctx->gn_counters.p, sizeof(c), hipMemcpyDeviceToHost));
  *count_out = c;  // counted only by calls made with "debug.ablate" bit 9 set
  return F360_OK;
}
