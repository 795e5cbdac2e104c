// image_sampler.hip -- ImageSampler device side: point-sampled log-rectilinear
// warp, log-polar forward warp, log-polar bilinear un-warp, 3x3 blur.
//
// Replaces ImageSampler::{InitializeGrid, InitializeLogpolarGrid,
// SampleFrameRectGPU, SampleFrameLogPolarGPU, InterpolateFrameLogPolarGPU,
// ApplyLogPolarGaussianBlur} (src/image_sampler.cc:170-299,577-621,780-857) and
// their kernels (src/image_sampler_sample_rect_kernel.cl,
// src/image_sampler_sample_logpolar_kernel.cl,
// src/image_sampler_interpolate_kernel.cl).
#include <cmath>

#include "f360_internal.h"
#include "host_tables.h"

namespace {

// OpenCL float builtins as correctly rounded floats (DESIGN.md "Float model").
__device__ __forceinline__ float cr_logf(float x) { return (float)log((double)x); }
__device__ __forceinline__ float cr_atanf(float x) { return (float)atan((double)x); }

__device__ __forceinline__ float mixf(float a, float b, float t) {
  return a + (b - a) * t;
}
typedef float f32x2_is __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(4))) uint2_a4 {  // an 8-byte load from a 4-byte aligned address
  uint32_t x, y;
  __device__ operator uint2() const { return make_uint2(x, y); }
};

// host side: may a launch use 4-byte accesses?
inline bool word_pixels(const void *dst, int dst_linesize, int dst_w, const void *src,
                        int src_linesize, int src_w) {
  return dst_linesize / dst_w == 4 && src_linesize / src_w == 4 && dst_linesize % 4 == 0 &&
         src_linesize % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(src) & 3) == 0;
}

// A thread of the two point samplers owns kPointRows vertically adjacent reduced pixels: their
// table loads and then their texel loads go out together (one pixel per thread is two dependent
// round trips in a very short wave).
constexpr int kPointRows = 4;

// Three bytes of a pixel; the fourth byte of the target is the caller's.  `words`: source and
// target are 4-byte aligned 4-byte pixels -> one load and two stores instead of three and three.
__device__ __forceinline__ uint32_t load_px(const uint8_t *s, bool words) {
  if (words) return *reinterpret_cast<const uint32_t *>(s);
  return (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
}
__device__ __forceinline__ void store_px(uint8_t *o, uint32_t v, bool words) {
  if (words) {
    *reinterpret_cast<uint16_t *>(o) = (uint16_t)v;
    o[2] = (uint8_t)(v >> 16);
    return;
  }
  o[0] = (uint8_t)v;
  o[1] = (uint8_t)(v >> 8);
  o[2] = (uint8_t)(v >> 16);
}

// sample_rect_kernel, src/image_sampler_sample_rect_kernel.cl:1-46
__global__ __launch_bounds__(256) void is_sample_rect_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_linesize, int obpp,
    const uint8_t *__restrict__ src, int src_w, int src_h, int src_linesize,
    int sbpp, const int16_t *__restrict__ gx, const int16_t *__restrict__ gy,
    float cxf, float cyf, bool words, bool bands) {
  int bx, by;
  f360::xcd_band_block(bands, bx, by);
  const int i = bx * 64 + (threadIdx.x & 63);
  const int j0 = (by * 4 + (threadIdx.x >> 6)) * kPointRows;
  if (i >= out_w || j0 >= out_h) return;
  int xp = (int)(cxf + (float)gx[i]);  // float add, then truncation (:24-25)
  if (xp >= src_w)
    xp -= src_w;
  else if (xp < 0)
    xp += src_w;
  int yp[kPointRows];
#pragma unroll
  for (int k = 0; k < kPointRows; ++k)
    yp[k] = (int)(cyf + (float)gy[min(j0 + k, out_h - 1)]);
  const bool x_ok = xp >= 0 && xp < src_w;
  uint32_t px[kPointRows];
#pragma unroll
  for (int k = 0; k < kPointRows; ++k)
    if (x_ok && yp[k] >= 0 && yp[k] < src_h)
      px[k] = load_px(src + (size_t)yp[k] * src_linesize + (size_t)xp * sbpp, words);
#pragma unroll
  for (int k = 0; k < kPointRows; ++k)
    if (j0 + k < out_h && x_ok && yp[k] >= 0 && yp[k] < src_h)
      store_px(dst + (size_t)(j0 + k) * out_linesize + (size_t)i * obpp, px[k], words);
}

// sample_logpolar_kernel, src/image_sampler_sample_logpolar_kernel.cl:41-86,
// with the grid of :5-39 recomputed from its separable factors:
// grid.x = (short)(int)(radius[i] * cos[j]), grid.y likewise with sin.
__global__ __launch_bounds__(256) void is_sample_logpolar_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_linesize, int obpp,
    const uint8_t *__restrict__ src, int src_w, int src_h, int src_linesize,
    int sbpp, const float *__restrict__ rad, const float *__restrict__ cs,
    const float *__restrict__ sn, float cxf, float cyf, bool words, bool bands) {
  int bx, by;
  f360::xcd_band_block(bands, bx, by);
  const int i = bx * 64 + (threadIdx.x & 63);
  const int j0 = (by * 4 + (threadIdx.x >> 6)) * kPointRows;
  if (i >= out_w || j0 >= out_h) return;
  const float r = rad[i];
  float cj[kPointRows], sj[kPointRows];
#pragma unroll
  for (int k = 0; k < kPointRows; ++k) {
    cj[k] = cs[min(j0 + k, out_h - 1)];
    sj[k] = sn[min(j0 + k, out_h - 1)];
  }
  uint32_t px[kPointRows];
  bool ok[kPointRows];
#pragma unroll
  for (int k = 0; k < kPointRows; ++k) {
    const int gxv = (int)(int16_t)(int)(r * cj[k]);
    const int gyv = (int)(int16_t)(int)(r * sj[k]);
    int xp = (int)(cxf + (float)gxv);
    int yp = (int)(cyf + (float)gyv);
    xp = (xp + 10 * src_w) % src_w;
    yp = min(max(yp, 0), src_h - 1);
    ok[k] = xp >= 0 && xp < src_w && yp >= 0 && yp < src_h;
    if (ok[k]) px[k] = load_px(src + (size_t)yp * src_linesize + (size_t)xp * sbpp, words);
  }
#pragma unroll
  for (int k = 0; k < kPointRows; ++k)
    if (j0 + k < out_h && ok[k])
      store_px(dst + (size_t)(j0 + k) * out_linesize + (size_t)i * obpp, px[k], words);
}

// interpolate_logpolar_kernel, src/image_sampler_interpolate_kernel.cl:1-81
//
// Step 1 of the kernel (:26-44), the reduced-buffer coordinates of an output pixel, depends on
// the INTEGER offset (delta_x, delta_y) from the truncated gaze centre and on the geometry only
// -- not on the gaze.  It is also where all the time goes (log, sqrt, atan, fmod, each evaluated
// in double and rounded once: 306 us per 8K frame, 0.06 of the HBM roofline).
__device__ __forceinline__ float2 logpolar_uv(int dx, int dy, int rw, int rh, int src_h) {
  float i_f = 0.0f;
  if (dx != 0 || dy != 0) {
    // pow(d, 2.0f) correctly rounded == one rounding of the exact product;
    // pow(v, 1.0f) == v
    const float fx = (float)dx, fy = (float)dy;
    const float r2 = fx * fx + fy * fy;
    i_f = (float)rw * (cr_logf(sqrtf(r2)) / 10.0f);
  }
  float j_f;
  if (dx != 0) {
    j_f = (float)(((double)cr_atanf((float)dy / (float)dx) +
                   M_PI * (double)(dx < 0)) *
                  ((double)(float)rh / (2.0 * M_PI)));
    j_f = fmodf(j_f + (float)(2 * rh), (float)src_h);
  } else {
    j_f = (float)((M_PI_2 + M_PI * (double)(dy < 0)) *
                  ((double)rh / (2.0 * M_PI)));
  }
  return make_float2(i_f, j_f);
}

// out of line: the un-warp kernel calls it only for offsets outside its table, and inlining the
// double-precision polynomials eight times costs it a third of its occupancy
__device__ __noinline__ float2 logpolar_uv_outlined(int dx, int dy, int rw, int rh, int src_h) {
  return logpolar_uv(dx, dy, rw, rh, src_h);
}

// Hence a table over the offsets, built once per geometry and shared by every gaze:
// uv[(dy + span_y) * pitch + (dx + span_x)], |dx| <= span_x = W/2 + 1, |dy| <= span_y = H
// (every offset a gaze inside the frame can produce; others are computed directly).
//
// Measured against this layout in round 2 and not kept (A/B on one box, 8K, this kernel's
// ~150 instructions per pixel and the way its loads batch across the four rows decide its time,
// not the table's bytes): a one-quadrant table of { i_f, atan(|dy| / |dx|) } with j_f rebuilt
// from the signs (118 instead of 472 MB; 151 -> 174 us); a flag in i_f's spare sign bit for
// "the exact-hit test cannot pass at this offset" (true for four pixels in five) that skips the
// radius / cos / sin loads (151 -> 178 us: loads inside a branch stop the compiler from
// batching them) or only the double arithmetic (151 -> 192 us).
struct LogpolarTable {
  const float2 *uv;
  int span_x, span_y, pitch;
};

__global__ __launch_bounds__(256) void logpolar_table_kernel(float2 *__restrict__ uv, int span_x,
                                                             int span_y, int pitch, int rw, int rh,
                                                             int src_h) {
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ix > 2 * span_x || iy > 2 * span_y) return;
  uv[(size_t)iy * pitch + ix] = logpolar_uv(ix - span_x, iy - span_y, rw, rh, src_h);
}

// A thread owns kLpRows vertically adjacent output pixels and runs the kernel's dependent steps
// (offset table -> radius / cos / sin tables -> texels) for all of them at once: four memory
// round trips per kLpRows pixels instead of per pixel (one pixel per thread was a chain of round
// trips in 460 k short-lived waves at 8K).
constexpr int kLpRows = 4;

// m % n for the kernel's (int)floor(j_float + source_height) % source_height: j_float is the
// result of an fmod by source_height (or a quarter / three quarters of it), so m lies in
// [n, 2n]; anything else takes the integer division
__device__ __forceinline__ int mod_near(int m, int n) {
  if (m >= 0 && m < 3 * n) return m - (m >= n ? n : 0) - (m >= 2 * n ? n : 0);
  return m % n;
}

__global__ __launch_bounds__(256) void is_interpolate_logpolar_kernel(
    uint32_t *__restrict__ dst, int out_w, int out_h,
    const uint32_t *__restrict__ src, int src_w, int src_h,
    const float *__restrict__ rad, const double *__restrict__ cs,
    const double *__restrict__ sn, float cxf, float cyf, int cxp, int cyp,
    const LogpolarTable table) {
  const int x0 = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kLpRows;
  if (x0 >= out_w || y0 >= out_h) return;
  const int rw = src_w, rh = src_h;
  int x = x0;
  if (x - cxp > out_w / 2)
    x -= out_w;
  else if (x - cxp < (-out_w) / 2)
    x += out_w;
  const int dx = x - cxp;

  float2 uv[kLpRows];
#pragma unroll
  for (int k = 0; k < kLpRows; ++k) {
    const int dy = min(y0 + k, out_h - 1) - cyp;
    if (table.uv && abs(dx) <= table.span_x && abs(dy) <= table.span_y)
      uv[k] = table.uv[(size_t)(dy + table.span_y) * table.pitch + (dx + table.span_x)];
    else
      uv[k] = logpolar_uv_outlined(dx, dy, rw, rh, src_h);
  }
  int ii[kLpRows], jj[kLpRows];
  float radius[kLpRows];
  double cj[kLpRows], sj[kLpRows];
#pragma unroll
  for (int k = 0; k < kLpRows; ++k) {
    ii[k] = min(max((int)roundf(uv[k].x), 0), rw - 1);
    jj[k] = min(max((int)roundf(uv[k].y), 0), rh - 1);
    radius[k] = rad[ii[k]];
    cj[k] = cs[jj[k]];
    sj[k] = sn[jj[k]];
  }
  bool exact[kLpRows];
  uint32_t tl[kLpRows], tr[kLpRows], bl[kLpRows], br[kLpRows];
#pragma unroll
  for (int k = 0; k < kLpRows; ++k) {
    const int y = min(y0 + k, out_h - 1);
    const int calc_x = (int)((double)cxf + (double)radius[k] * cj[k]);
    const int calc_y = (int)((double)cyf + (double)radius[k] * sj[k]);
    exact[k] = calc_x == x && calc_y == y;
    if (exact[k]) {
      tl[k] = src[(size_t)jj[k] * src_w + ii[k]];
    } else {
      const float i_f = uv[k].x, j_f = uv[k].y;
      const int min_i = min(max((int)floorf(i_f), 0), src_w - 1);
      const int min_j = mod_near((int)floorf(j_f + (float)src_h), src_h);
      const int max_i = min(max((int)ceilf(i_f), 0), src_w - 1);
      const int max_j = mod_near((int)ceilf(j_f + (float)src_h), src_h);
      tl[k] = src[(size_t)min_j * src_w + min_i];
      tr[k] = src[(size_t)min_j * src_w + max_i];
      bl[k] = src[(size_t)max_j * src_w + min_i];
      br[k] = src[(size_t)max_j * src_w + max_i];
    }
  }
#pragma unroll
  for (int k = 0; k < kLpRows; ++k) {
    if (y0 + k >= out_h) break;
    uint32_t out;
    if (exact[k]) {
      out = tl[k] & 0x00ffffffu;
    } else {
      const float i_f = uv[k].x, j_f = uv[k].y;
      const float ir = i_f - floorf(i_f), jr = j_f - floorf(j_f);
      out = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float l = mixf((float)((tl[k] >> (8 * c)) & 0xffu), (float)((bl[k] >> (8 * c)) & 0xffu), jr);
        const float r = mixf((float)((tr[k] >> (8 * c)) & 0xffu), (float)((br[k] >> (8 * c)) & 0xffu), jr);
        out |= ((uint32_t)(int)mixf(l, r, ir) & 0xffu) << (8 * c);
      }
    }
    dst[(size_t)(y0 + k) * out_w + x0] = out;
  }
}

// The same kernel for the common case -- the whole frame inside the offset table, the three axis
// tables small enough for LDS -- restructured around what bounded the kernel above (1109
// instructions for four pixels, 33 branches, 22 waits; 7 scattered requests per pixel):
//   * the radius / cos / sin tables of the exact-hit test (:47-52) sit in LDS (17 + 34 KB at 8K),
//     loaded once by a workgroup that then loops over tiles: three global gathers per pixel
//     become one 4-byte and one 16-byte LDS read;
//   * the four texels are two 8-byte loads: ceil(i) is floor(i) or floor(i) + 1, so the left and
//     right neighbour are adjacent in their row (the pair starts at min(min_i, width - 2));
//   * the exact-hit texel (j, i) is one of those four unless float rounding in `j_float +
//     source_height` or the clamp of j at the last row says otherwise: a wave-uniform, almost
//     never taken branch fetches it then;
//   * no branch on the main path (floor / ceil / round share one v_floor; the `%` of :59-61 is two
//     conditional subtractions: its argument lies in [n, 2n + 1]), so the compiler issues the
//     loads of a thread's four rows together: three round trips per four pixels;
//   * the lerps run two values per packed-float instruction, same operations in the same order.
// i_float and j_float are never negative (log of a radius >= 1; fmod of a positive number; a
// positive constant), which round() as floor + (fraction >= 0.5) relies on.
constexpr uint32_t kLpLdsMax = 96 * 1024;
constexpr int kLpLdsRows = 2;  // output rows per thread (8K: 148 us with 8, 130 with 4, 122 with 2, 137 with 1)

__device__ __forceinline__ f32x2_is lerp2(f32x2_is a, f32x2_is b, float t) {
  return a + (b - a) * t;  // -ffp-contract=off: sub, mul, add like mix()
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void is_interpolate_logpolar_lds_kernel(
    uint32_t *__restrict__ dst, int out_w, int out_h, const uint32_t *__restrict__ src, int src_w,
    int src_h, const float *__restrict__ rad, const double *__restrict__ cs,
    const double *__restrict__ sn, float cxf, float cyf, int cxp, int cyp,
    const LogpolarTable table, int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lp_lds[];
  double2 *l_cs = reinterpret_cast<double2 *>(lp_lds);           // [src_h] {cos, sin}
  float *l_rad = reinterpret_cast<float *>(l_cs + src_h);        // [src_w]
  for (int i = threadIdx.x; i < src_h; i += THREADS) l_cs[i] = make_double2(cs[i], sn[i]);
  for (int i = threadIdx.x; i < src_w; i += THREADS) l_rad[i] = rad[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rw = src_w, rh = src_h;
  const double cxd = (double)cxf, cyd = (double)cyf;
  const float hf = (float)src_h;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * 64 + lane;
    const int y0 = (ty * (THREADS / 64) + wave) * kLpLdsRows;
    if (x0 >= out_w || y0 >= out_h) continue;
    int x = x0;
    if (x - cxp > out_w / 2)
      x -= out_w;
    else if (x - cxp < (-out_w) / 2)
      x += out_w;
    const int dx = x - cxp;
    const uint32_t col = (uint32_t)min(max(dx + table.span_x, 0), 2 * table.span_x);

    float2 uv[kLpLdsRows];
#pragma unroll
    for (int k = 0; k < kLpLdsRows; ++k) {
      const int dy = min(y0 + k, out_h - 1) - cyp;
      const uint32_t row = (uint32_t)min(max(dy + table.span_y, 0), 2 * table.span_y);
      // (24-bit multiplies are full rate, 32-bit ones quarter rate: rows, pitches and widths
      // are below 2^17 + 4, the products below 2^29 -- host check)
      uv[k] = table.uv[__umul24(row, (uint32_t)table.pitch) + col];
    }
    int ii[kLpLdsRows], jj[kLpLdsRows], min_i[kLpLdsRows], max_i[kLpLdsRows], min_j[kLpLdsRows], max_j[kLpLdsRows];
    float ir[kLpLdsRows], jr[kLpLdsRows];
    bool exact[kLpLdsRows];
    uint2 top[kLpLdsRows], bot[kLpLdsRows];
    int base_i[kLpLdsRows];
#pragma unroll
    for (int k = 0; k < kLpLdsRows; ++k) {
      const float i_f = uv[k].x, j_f = uv[k].y;
      const float fi = floorf(i_f), fj = floorf(j_f);
      ir[k] = i_f - fi;
      jr[k] = j_f - fj;
      const int i0 = (int)fi, j0 = (int)fj;
      // (unsigned min: the lower clamp of :34,44,57,59 cannot act on a non-negative value, and a
      // wild table entry still lands inside the buffer)
      ii[k] = (int)min((uint32_t)(i0 + (ir[k] >= 0.5f ? 1 : 0)), (uint32_t)(rw - 1));
      jj[k] = (int)min((uint32_t)(j0 + (jr[k] >= 0.5f ? 1 : 0)), (uint32_t)(rh - 1));
      min_i[k] = (int)min((uint32_t)i0, (uint32_t)(rw - 1));
      max_i[k] = (int)min((uint32_t)(i0 + (ir[k] > 0.0f ? 1 : 0)), (uint32_t)(rw - 1));
      const float t = j_f + hf, ft = floorf(t);
      // m % n for m in [n, 3n): m - n or m - 2n, whichever does not wrap below zero
      const uint32_t m0 = (uint32_t)(int)ft - (uint32_t)rh, m1 = m0 + (t > ft ? 1u : 0u);
      min_j[k] = (int)min(min(m0, m0 - (uint32_t)rh), (uint32_t)(rh - 1));
      max_j[k] = (int)min(min(m1, m1 - (uint32_t)rh), (uint32_t)(rh - 1));
      // (the last min only keeps a wild table entry inside the buffer)
      base_i[k] = min(min_i[k], rw - 2);
      top[k] = *reinterpret_cast<const uint2_a4 *>(
          src + (__umul24((uint32_t)min_j[k], (uint32_t)rw) + (uint32_t)base_i[k]));
      bot[k] = *reinterpret_cast<const uint2_a4 *>(
          src + (__umul24((uint32_t)max_j[k], (uint32_t)rw) + (uint32_t)base_i[k]));
    }
#pragma unroll
    for (int k = 0; k < kLpLdsRows; ++k) {
      const int y = min(y0 + k, out_h - 1);
      const float radius = l_rad[ii[k]];
      const double2 c = l_cs[jj[k]];
      const int calc_x = (int)(cxd + (double)radius * c.x);
      const int calc_y = (int)(cyd + (double)radius * c.y);
      exact[k] = calc_x == x && calc_y == y;
    }
    uint32_t out[kLpLdsRows];
    bool any_exact = false;
#pragma unroll
    for (int k = 0; k < kLpLdsRows; ++k) {
      const bool lo = min_i[k] == base_i[k], hi = max_i[k] == base_i[k];
      const uint32_t tl = lo ? top[k].x : top[k].y, tr = hi ? top[k].x : top[k].y;
      const uint32_t bl = lo ? bot[k].x : bot[k].y, br = hi ? bot[k].x : bot[k].y;
      const f32x2_is t01 = {(float)(tl & 0xffu), (float)((tl >> 8) & 0xffu)};
      const f32x2_is b01 = {(float)(bl & 0xffu), (float)((bl >> 8) & 0xffu)};
      const f32x2_is u01 = {(float)(tr & 0xffu), (float)((tr >> 8) & 0xffu)};
      const f32x2_is c01 = {(float)(br & 0xffu), (float)((br >> 8) & 0xffu)};
      const f32x2_is t2 = {(float)((tl >> 16) & 0xffu), (float)((tr >> 16) & 0xffu)};
      const f32x2_is b2 = {(float)((bl >> 16) & 0xffu), (float)((br >> 16) & 0xffu)};
      const f32x2_is l01 = lerp2(t01, b01, jr[k]);  // left colour, channels 0 and 1
      const f32x2_is r01 = lerp2(u01, c01, jr[k]);  // right colour, channels 0 and 1
      const f32x2_is v2 = lerp2(t2, b2, jr[k]);     // channel 2: {left, right}
      const f32x2_is h01 = lerp2(l01, r01, ir[k]);
      const float h2 = mixf(v2.x, v2.y, ir[k]);
      out[k] = ((uint32_t)(int)h01.x & 0xffu) | (((uint32_t)(int)h01.y & 0xffu) << 8) |
               (((uint32_t)(int)h2 & 0xffu) << 16);
      any_exact = any_exact | exact[k];
    }
    // exact hits (:53-55) are few and sit around the fovea: most waves have none.  The texel
    // (j, i) is one of the four already here unless float rounding in `j_float + source_height`
    // or the clamp of j at the last row says otherwise; then it is fetched.
    if (__any(any_exact)) {
#pragma unroll
      for (int k = 0; k < kLpLdsRows; ++k) {
        const bool lo = min_i[k] == base_i[k], hi = max_i[k] == base_i[k];
        const uint32_t tl = lo ? top[k].x : top[k].y, tr = hi ? top[k].x : top[k].y;
        const uint32_t bl = lo ? bot[k].x : bot[k].y, br = hi ? bot[k].x : bot[k].y;
        const bool i_lo = ii[k] == min_i[k], j_lo = jj[k] == min_j[k];
        const bool near_hit = (i_lo | (ii[k] == max_i[k])) & (j_lo | (jj[k] == max_j[k]));
        uint32_t hit = j_lo ? (i_lo ? tl : tr) : (i_lo ? bl : br);
        if (exact[k] & !near_hit)
          hit = src[__umul24((uint32_t)jj[k], (uint32_t)rw) + (uint32_t)ii[k]];
        if (exact[k]) out[k] = hit & 0x00ffffffu;
      }
    }
#pragma unroll
    for (int k = 0; k < kLpLdsRows; ++k)
      if (y0 + k < out_h) dst[__umul24((uint32_t)(y0 + k), (uint32_t)out_w) + (uint32_t)x0] = out[k];
  }
}

// logpolar_gaussian_blur_kernel, src/image_sampler_sample_logpolar_kernel.cl:88-142
__global__ __launch_bounds__(256) void is_blur_kernel(uint32_t *__restrict__ dst,
                                                      int w, int h,
                                                      const uint32_t *__restrict__ src) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= w || j >= h) return;
  if (i < w / 2) {
    dst[(size_t)j * w + i] = src[(size_t)j * w + i] & 0x00ffffffu;
    return;
  }
  const float P1 = (float)0.3377, P2 = (float)0.1217, P3 = (float)0.0439;
  const int jm = max(j - 1, 0), jp = min(j + 1, h - 1);
  const int im = max(i - 1, 0), ip = min(i + 1, w - 1);
  const uint32_t t11 = src[(size_t)jm * w + im], t12 = src[(size_t)jm * w + i],
                 t13 = src[(size_t)jm * w + ip], t21 = src[(size_t)j * w + im],
                 t22 = src[(size_t)j * w + i], t23 = src[(size_t)j * w + ip],
                 t31 = src[(size_t)jp * w + im], t32 = src[(size_t)jp * w + i],
                 t33 = src[(size_t)jp * w + ip];
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#define F360_CH(t) ((float)(((t) >> (8 * c)) & 0xffu))
    const float corners = F360_CH(t11) + F360_CH(t13) + F360_CH(t31) + F360_CH(t33);
    const float edges = F360_CH(t12) + F360_CH(t21) + F360_CH(t23) + F360_CH(t32);
    const float v = P3 * corners + P2 * edges + P1 * F360_CH(t22);
#undef F360_CH
    out |= ((uint32_t)(int)v & 0xffu) << (8 * c);
  }
  dst[(size_t)j * w + i] = out;
}

int upload(f360_ctx *ctx, f360::DevBuf &buf, const void *host, size_t bytes) {
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  int st = buf.reserve(bytes);
  if (st != F360_OK) return st;
  F360_HIP_TRY(hipMemcpy(buf.p, host, bytes, hipMemcpyHostToDevice));
  return F360_OK;
}

bool bad_centre(float c) { return !(std::fabs(c) <= 16.0f); }

}  // namespace

extern "C" {

int f360_is_create(f360_ctx *ctx, f360_image_sampler **out) {
  F360_REQUIRE(ctx && out, "f360_is_create: null argument");
  f360_image_sampler *s = new f360_image_sampler();
  s->ctx = ctx;
  *out = s;
  return F360_OK;
}

int f360_is_destroy(f360_image_sampler *is) {
  if (!is) return F360_OK;
  (void)hipStreamSynchronize(is->ctx->stream);
  is->gx_dev.release();
  is->gy_dev.release();
  is->lrad_dev.release();
  is->lcos_dev.release();
  is->lsin_dev.release();
  is->irad_dev.release();
  is->icos_dev.release();
  is->isin_dev.release();
  is->lpt_dev.release();
  delete is;
  return F360_OK;
}

int f360_is_initialize_grid(f360_image_sampler *is, int target_width,
                            int target_height, int source_width,
                            int source_height) {
  F360_REQUIRE(is, "f360_is_initialize_grid: null sampler");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 1 &&
                   source_height >= 1,
               "f360_is_initialize_grid: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_is_initialize_grid: a dimension exceeds 65536");
  if (is->gw == target_width && is->gh == target_height && is->sw == source_width &&
      is->sh == source_height && is->gx_dev.p)
    return F360_OK;
  F360_BIND_DEVICE(is->ctx);
  f360::build_is_grid_axis(is->gx_host, target_width, source_width);
  f360::build_is_grid_axis(is->gy_host, target_height, source_height);
  int st = upload(is->ctx, is->gx_dev, is->gx_host.data(),
                  is->gx_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  st = upload(is->ctx, is->gy_dev, is->gy_host.data(),
              is->gy_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  is->gw = target_width;
  is->gh = target_height;
  is->sw = source_width;
  is->sh = source_height;
  return F360_OK;
}

int f360_is_initialize_logpolar_grid(f360_image_sampler *is, int target_width,
                                     int target_height, int source_width,
                                     int source_height) {
  F360_REQUIRE(is, "f360_is_initialize_logpolar_grid: null sampler");
  F360_REQUIRE(target_width >= 1 && target_height >= 1,
               "f360_is_initialize_logpolar_grid: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_is_initialize_logpolar_grid: a dimension exceeds 65536");
  if (is->lw == target_width && is->lh == target_height && is->lrad_dev.p) {
    is->lsw = source_width;
    is->lsh = source_height;
    return F360_OK;
  }
  F360_BIND_DEVICE(is->ctx);
  f360::build_logpolar_axes(is->lrad_host, is->lcos_host, is->lsin_host,
                            target_width, target_height);
  int st = upload(is->ctx, is->lrad_dev, is->lrad_host.data(),
                  is->lrad_host.size() * sizeof(float));
  if (st != F360_OK) return st;
  st = upload(is->ctx, is->lcos_dev, is->lcos_host.data(),
              is->lcos_host.size() * sizeof(float));
  if (st != F360_OK) return st;
  st = upload(is->ctx, is->lsin_dev, is->lsin_host.data(),
              is->lsin_host.size() * sizeof(float));
  if (st != F360_OK) return st;
  is->lw = target_width;
  is->lh = target_height;
  is->lsw = source_width;
  is->lsh = source_height;
  return F360_OK;
}

int f360_is_export_grid(f360_image_sampler *is, int16_t *grid_host) {
  F360_REQUIRE(is && grid_host, "f360_is_export_grid: null argument");
  if (!is->gx_dev.p) {
    f360::set_error("f360_is_export_grid: grid not initialised");
    return F360_ERR_NOT_INITIALIZED;
  }
  std::vector<int16_t> gx(is->gx_host.size()), gy(is->gy_host.size());
  F360_HIP_TRY(hipStreamSynchronize(is->ctx->stream));
  F360_HIP_TRY(hipMemcpy(gx.data(), is->gx_dev.p, gx.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  F360_HIP_TRY(hipMemcpy(gy.data(), is->gy_dev.p, gy.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  for (int j = 0; j < is->gh; ++j)
    for (int i = 0; i < is->gw; ++i) {
      grid_host[((size_t)j * is->gw + i) * 2 + 0] = gx[(size_t)i];
      grid_host[((size_t)j * is->gw + i) * 2 + 1] = gy[(size_t)j];
    }
  return F360_OK;
}

int f360_is_export_logpolar_grid(f360_image_sampler *is, int16_t *grid_host) {
  F360_REQUIRE(is && grid_host, "f360_is_export_logpolar_grid: null argument");
  if (!is->lrad_dev.p) {
    f360::set_error("f360_is_export_logpolar_grid: grid not initialised");
    return F360_ERR_NOT_INITIALIZED;
  }
  std::vector<float> rad(is->lrad_host.size()), cs(is->lcos_host.size()),
      sn(is->lsin_host.size());
  F360_HIP_TRY(hipStreamSynchronize(is->ctx->stream));
  F360_HIP_TRY(hipMemcpy(rad.data(), is->lrad_dev.p, rad.size() * sizeof(float),
                         hipMemcpyDeviceToHost));
  F360_HIP_TRY(hipMemcpy(cs.data(), is->lcos_dev.p, cs.size() * sizeof(float),
                         hipMemcpyDeviceToHost));
  F360_HIP_TRY(hipMemcpy(sn.data(), is->lsin_dev.p, sn.size() * sizeof(float),
                         hipMemcpyDeviceToHost));
  for (int j = 0; j < is->lh; ++j)
    for (int i = 0; i < is->lw; ++i) {
      grid_host[((size_t)j * is->lw + i) * 2 + 0] =
          (int16_t)(int)(rad[(size_t)i] * cs[(size_t)j]);
      grid_host[((size_t)j * is->lw + i) * 2 + 1] =
          (int16_t)(int)(rad[(size_t)i] * sn[(size_t)j]);
    }
  return F360_OK;
}

int f360_is_sample_rect(f360_image_sampler *is, uint8_t *target_dev,
                        int target_width, int target_height,
                        int target_linesize, const uint8_t *source_dev,
                        int source_width, int source_height,
                        int source_linesize, float center_x, float center_y) {
  F360_REQUIRE(is, "f360_is_sample_rect: null sampler");
  F360_BIND_DEVICE(is->ctx);
  F360_REQUIRE(target_dev && source_dev, "f360_is_sample_rect: null buffer");
  if (!is->gx_dev.p) {  // the reference never auto-initialises (image_sampler.cc:261)
    f360::set_error("f360_is_sample_rect: InitializeGrid has not been called");
    return F360_ERR_NOT_INITIALIZED;
  }
  F360_REQUIRE(is->gw == target_width && is->gh == target_height,
               "f360_is_sample_rect: grid was initialised for %dx%d", is->gw, is->gh);
  F360_REQUIRE(target_linesize / target_width >= 3 &&
                   source_linesize / source_width >= 3,
               "f360_is_sample_rect: need >= 3 bytes per pixel");
  F360_REQUIRE(!bad_centre(center_x) && !bad_centre(center_y),
               "f360_is_sample_rect: gaze centre out of range");
  const dim3 grid((target_width + 63) / 64, (target_height + 4 * kPointRows - 1) / (4 * kPointRows));
  hipLaunchKernelGGL(is_sample_rect_kernel, grid, dim3(256), 0, is->ctx->stream,
                     target_dev, target_width, target_height, target_linesize,
                     target_linesize / target_width, source_dev, source_width,
                     source_height, source_linesize, source_linesize / source_width,
                     is->gx_dev.as<int16_t>(), is->gy_dev.as<int16_t>(),
                     center_x * (float)source_width,
                     center_y * (float)source_height,
                     word_pixels(target_dev, target_linesize, target_width, source_dev,
                                 source_linesize, source_width),
                     // row bands pay where the source is far beyond the L2s (8K: 26.5 -> 20.7 us);
                     // at 3840x1920 and below the round-robin order's balance is worth more
                     is->ctx->opt_xcd_bands == 2 ||
                         (is->ctx->opt_xcd_bands == 1 &&
                          (size_t)source_linesize * source_height >= ((size_t)64 << 20)));
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_is_sample_logpolar(f360_image_sampler *is, uint8_t *target_dev,
                            int target_width, int target_height,
                            int target_linesize, const uint8_t *source_dev,
                            int source_width, int source_height,
                            int source_linesize, float center_x,
                            float center_y) {
  F360_REQUIRE(is, "f360_is_sample_logpolar: null sampler");
  F360_BIND_DEVICE(is->ctx);
  F360_REQUIRE(target_dev && source_dev, "f360_is_sample_logpolar: null buffer");
  if (!is->lrad_dev.p) {
    f360::set_error(
        "f360_is_sample_logpolar: InitializeLogpolarGrid has not been called");
    return F360_ERR_NOT_INITIALIZED;
  }
  F360_REQUIRE(is->lw == target_width && is->lh == target_height,
               "f360_is_sample_logpolar: grid was initialised for %dx%d", is->lw,
               is->lh);
  F360_REQUIRE(target_linesize / target_width >= 3 &&
                   source_linesize / source_width >= 3,
               "f360_is_sample_logpolar: need >= 3 bytes per pixel");
  F360_REQUIRE(!bad_centre(center_x) && !bad_centre(center_y),
               "f360_is_sample_logpolar: gaze centre out of range");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}),
               "f360_is_sample_logpolar: a dimension exceeds 65536");
  const dim3 grid((target_width + 63) / 64, (target_height + 4 * kPointRows - 1) / (4 * kPointRows));
  hipLaunchKernelGGL(is_sample_logpolar_kernel, grid, dim3(256), 0, is->ctx->stream,
                     target_dev, target_width, target_height, target_linesize,
                     target_linesize / target_width, source_dev, source_width,
                     source_height, source_linesize, source_linesize / source_width,
                     is->lrad_dev.as<float>(), is->lcos_dev.as<float>(),
                     is->lsin_dev.as<float>(), center_x * (float)source_width,
                     center_y * (float)source_height,
                     word_pixels(target_dev, target_linesize, target_width, source_dev,
                                 source_linesize, source_width),
                     // (rays, not rows: 31.5 -> 30.4 us at 8K, slower below; only when forced)
                     is->ctx->opt_xcd_bands == 2);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_is_interpolate_logpolar(f360_image_sampler *is, uint8_t *target_dev,
                                 int target_width, int target_height,
                                 int target_linesize, const uint8_t *source_dev,
                                 int source_width, int source_height,
                                 int source_linesize, float center_x,
                                 float center_y) {
  (void)target_linesize;
  (void)source_linesize;
  F360_REQUIRE(is, "f360_is_interpolate_logpolar: null sampler");
  F360_BIND_DEVICE(is->ctx);
  F360_REQUIRE(target_dev && source_dev, "f360_is_interpolate_logpolar: null buffer");
  F360_REQUIRE(target_width >= 2 && target_height >= 2 && source_width >= 1 &&
                   source_height >= 1,
               "f360_is_interpolate_logpolar: bad geometry");
  F360_REQUIRE(f360::dims_ok({target_width, target_height, source_width, source_height}), "f360_is_interpolate_logpolar: a dimension exceeds 65536");
  F360_REQUIRE(((uintptr_t)target_dev % 4) == 0 && ((uintptr_t)source_dev % 4) == 0,
               "f360_is_interpolate_logpolar: buffers must be 4-byte aligned");
  F360_REQUIRE(!bad_centre(center_x) && !bad_centre(center_y),
               "f360_is_interpolate_logpolar: gaze centre out of range");
  if (is->iw != source_width || is->ih != source_height || !is->irad_dev.p) {
    std::vector<float> rad;
    std::vector<double> cs, sn;
    f360::build_logpolar_inverse_axes(rad, cs, sn, source_width, source_height);
    int st = upload(is->ctx, is->irad_dev, rad.data(), rad.size() * sizeof(float));
    if (st != F360_OK) return st;
    st = upload(is->ctx, is->icos_dev, cs.data(), cs.size() * sizeof(double));
    if (st != F360_OK) return st;
    st = upload(is->ctx, is->isin_dev, sn.data(), sn.size() * sizeof(double));
    if (st != F360_OK) return st;
    is->iw = source_width;
    is->ih = source_height;
  }
  // the offset -> (i, j) table of this geometry ("is.lp_table", up to 1 GiB; 472 MB at 8K)
  LogpolarTable table{nullptr, target_width / 2 + 1, target_height, 0};
  table.pitch = 2 * table.span_x + 1;
  const size_t table_bytes = (size_t)(2 * table.span_y + 1) * table.pitch * sizeof(float2);
  if (is->ctx->opt_lp_table && table_bytes <= ((size_t)1 << 30)) {
    if (is->lpt_w != target_width || is->lpt_h != target_height || is->lpt_sw != source_width ||
        is->lpt_sh != source_height || !is->lpt_dev.p) {
      F360_HIP_TRY(hipStreamSynchronize(is->ctx->stream));  // earlier calls may read the old one
      int st = is->lpt_dev.reserve(table_bytes);
      if (st != F360_OK) return st;
      const dim3 tgrid((table.pitch + 63) / 64, (2 * table.span_y + 1 + 3) / 4);
      hipLaunchKernelGGL(logpolar_table_kernel, tgrid, dim3(256), 0, is->ctx->stream,
                         is->lpt_dev.as<float2>(), table.span_x, table.span_y, table.pitch,
                         source_width, source_height, source_height);
      F360_HIP_TRY(hipGetLastError());
      is->lpt_w = target_width;
      is->lpt_h = target_height;
      is->lpt_sw = source_width;
      is->lpt_sh = source_height;
    }
    table.uv = is->lpt_dev.as<float2>();
  }
  const float cxf = center_x * (float)target_width;
  const float cyf = center_y * (float)target_height;
  const dim3 grid((target_width + 63) / 64, (target_height + 4 * kLpRows - 1) / (4 * kLpRows));
  f360::KernelSpan span(is->ctx, f360::kIsInterpolateLogpolar,
                        f360::take_profile_slot(is->ctx));
  // "is.lp_lds": the branch-free kernel with the axis tables in LDS -- whenever the gaze lies in
  // the frame (every offset is in the table then), the tables fit and 32-bit texel offsets do
  const size_t lds_bytes = (size_t)source_height * 16 + (size_t)source_width * 4;
  const int cxp = (int)cxf, cyp = (int)cyf;
  if (is->ctx->opt_lp_lds && table.uv && cxp >= 0 && cxp <= target_width && cyp >= 0 &&
      cyp <= target_height && source_width >= 2 && lds_bytes <= kLpLdsMax &&
      (size_t)source_width * source_height < ((size_t)1 << 29) &&
      (size_t)target_width * target_height < ((size_t)1 << 30) &&
      table_bytes / sizeof(float2) < ((size_t)1 << 29)) {
    if (!is->lp_lds_ready) {
      F360_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void *>(is_interpolate_logpolar_lds_kernel<256>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLpLdsMax));
      F360_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void *>(is_interpolate_logpolar_lds_kernel<512>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLpLdsMax));
      F360_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void *>(is_interpolate_logpolar_lds_kernel<1024>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLpLdsMax));
      int dev = 0, cus = 0;
      F360_HIP_TRY(hipGetDevice(&dev));
      F360_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      is->lp_cus = cus > 0 ? cus : 256;
      is->lp_lds_ready = true;
    }
    // workgroup size: the tables are per workgroup, so large tables want many waves behind them
    // (with two rows per thread: 8K 153 / 122 / 120 us for 256 / 512 / 1024 threads, 3840x1920
    // 33.6 / 31.5 / 31.4, 1920x1080 12.8 / 11.9 / 11.8 -- 512 everywhere)
    const int threads = is->ctx->opt_lp_lds > 1 ? is->ctx->opt_lp_lds : 512;
    const int rows_per_wg = threads / 64 * kLpLdsRows;
    const int tiles_x = (target_width + 63) / 64;
    const int ntiles = tiles_x * ((target_height + rows_per_wg - 1) / rows_per_wg);
    const int per_cu = (int)std::min<size_t>(
        2048 / threads, std::max<size_t>(1, (160 * 1024) / (lds_bytes + 512)));
    const int nwg = std::min(ntiles, is->lp_cus * per_cu);
#define F360_LP_LAUNCH(T)                                                                          \
  hipLaunchKernelGGL(is_interpolate_logpolar_lds_kernel<T>, dim3(nwg), dim3(T), lds_bytes,         \
                     is->ctx->stream, reinterpret_cast<uint32_t *>(target_dev), target_width,      \
                     target_height, reinterpret_cast<const uint32_t *>(source_dev), source_width,  \
                     source_height, is->irad_dev.as<float>(), is->icos_dev.as<double>(),           \
                     is->isin_dev.as<double>(), cxf, cyf, cxp, cyp, table, tiles_x, ntiles)
    if (threads == 512)
      F360_LP_LAUNCH(512);
    else if (threads == 1024)
      F360_LP_LAUNCH(1024);
    else
      F360_LP_LAUNCH(256);
#undef F360_LP_LAUNCH
    F360_HIP_TRY(hipGetLastError());
    return F360_OK;
  }
  hipLaunchKernelGGL(is_interpolate_logpolar_kernel, grid, dim3(256), 0,
                     is->ctx->stream, reinterpret_cast<uint32_t *>(target_dev),
                     target_width, target_height,
                     reinterpret_cast<const uint32_t *>(source_dev), source_width,
                     source_height, is->irad_dev.as<float>(),
                     is->icos_dev.as<double>(), is->isin_dev.as<double>(), cxf, cyf,
                     (int)cxf, (int)cyf, table);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_is_logpolar_gaussian_blur(f360_image_sampler *is, uint8_t *target_dev,
                                   int target_width, int target_height,
                                   int target_linesize,
                                   const uint8_t *source_dev) {
  (void)target_linesize;  // kernel indexes 4-byte texels with a row stride of width
  F360_REQUIRE(is, "f360_is_logpolar_gaussian_blur: null sampler");
  F360_BIND_DEVICE(is->ctx);
  F360_REQUIRE(target_dev && source_dev, "f360_is_logpolar_gaussian_blur: null buffer");
  F360_REQUIRE(target_width >= 1 && target_height >= 1,
               "f360_is_logpolar_gaussian_blur: bad geometry");
  F360_REQUIRE(((uintptr_t)target_dev % 4) == 0 && ((uintptr_t)source_dev % 4) == 0,
               "f360_is_logpolar_gaussian_blur: buffers must be 4-byte aligned");
  const dim3 grid((target_width + 63) / 64, (target_height + 3) / 4);
  hipLaunchKernelGGL(is_blur_kernel, grid, dim3(256), 0, is->ctx->stream,
                     reinterpret_cast<uint32_t *>(target_dev), target_width,
                     target_height, reinterpret_cast<const uint32_t *>(source_dev));
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // extern "C"
