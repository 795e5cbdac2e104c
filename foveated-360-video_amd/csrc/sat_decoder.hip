// sat_decoder.hip -- SATDecoder device side: log-rectilinear SAT sampler
// (forward warp), bilinear un-warp, identity decode.
//
// Replaces SATDecoder::{InitializeGrid, SampleFrameRectGPU,
// InterpolateFrameRectGPU, DecodeFrameGPU} (src/sat_decoder.cc:139-210,301-348,
// 887-928) and their kernels (src/sat_decoder_sample_rect_kernel.cl:138-295,
// src/sat_decoder_interpolate_kernel.cl, src/sat_decoder_decode_kernel.cl).
//
// The reference's 2-D grid and all of its per-pixel exp/pow/log calls are
// separable (x depends on the column only, y on the row only); they live in
// 1-D tables built on the host (host_tables.cpp) so the kernels below are pure
// integer / IEEE-float gather kernels.
#include <cmath>
#include <cstring>

#include "f360_internal.h"
#include "host_tables.h"

namespace {

// ---------------------------------------------------------------------------
// One axis of sample_rect_kernel (src/sat_decoder_sample_rect_kernel.cl
// :168-204): box corner `hi`, its lower partner `lo`, and whether the pixel is
// processed at all as far as this axis is concerned.
struct AxisBox {
  int hi, lo;
  bool ok;
};

__device__ __forceinline__ AxisBox sample_axis(int centre, int d_hi, int d_lo,
                                               int size, bool wraps) {
  int hi = centre + d_hi, lo = centre + d_lo;
  if (wraps) {  // only x wraps (:181-187); the y wrap is commented out
    if (hi >= size && lo >= size) {
      hi -= size;
      lo -= size;
    } else if (hi < 0 && lo < 0) {
      hi += size;
      lo += size;
    }
  }
  AxisBox b;
  b.ok = (hi >= 0 && hi < size) || (lo >= 0 && lo < size);
  b.hi = min(max(hi, 1), size - 1);
  b.lo = min(max(lo, 0), b.hi - 1);
  return b;
}

__device__ __forceinline__ uint3 load_sat3(const uint32_t *sat, size_t texel) {
  const uint32_t *p = sat + texel * 3;
  return make_uint3(p[0], p[1], p[2]);
}

// q = n / d for uint32, exact.  The common case (n < 2^23: a box of 8-bit
// samples) goes through one correctly rounded float division, which is exact
// there; anything larger (a wrapped table read with a degenerate box) takes
// the integer path.
__device__ __forceinline__ uint32_t udiv_exact(uint32_t n, uint32_t d) {
  if (n < (1u << 23) && d < (1u << 23)) {
    return (uint32_t)((float)n / (float)d);
  }
  return n / d;
}

// Variant 0: one thread per reduced pixel.
__global__ __launch_bounds__(256) void sample_rect_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_stride_px,
    const uint32_t *__restrict__ sat, int src_w, int src_h,
    const int16_t *__restrict__ gx, const int16_t *__restrict__ gy, int cxp,
    int cyp) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= out_w || j >= out_h) return;
  const AxisBox bx = sample_axis(cxp, gx[i + 1], gx[i], src_w, true);
  const AxisBox by = sample_axis(cyp, gy[j + 1], gy[j], src_h, false);
  if (!(bx.ok && by.ok)) return;
  const uint3 br = load_sat3(sat, (size_t)by.hi * src_w + bx.hi);
  const uint3 tr = load_sat3(sat, (size_t)by.lo * src_w + bx.hi);
  const uint3 tl = load_sat3(sat, (size_t)by.lo * src_w + bx.lo);
  const uint3 bl = load_sat3(sat, (size_t)by.hi * src_w + bx.lo);
  const uint32_t area = (uint32_t)((bx.hi - bx.lo) * (by.hi - by.lo));
  uint8_t *o = dst + ((size_t)j * out_stride_px + i) * 4;
  o[0] = (uint8_t)udiv_exact(br.x - tr.x + tl.x - bl.x, area);
  o[1] = (uint8_t)udiv_exact(br.y - tr.y + tl.y - bl.y, area);
  o[2] = (uint8_t)udiv_exact(br.z - tr.z + tl.z - bl.z, area);
}

// Variant 1 ("column walker"): a wave owns 63 adjacent reduced columns (lane 0
// is a halo lane for the column to the left) and walks down `rows` reduced rows.
// Adjacent boxes share corners -- the lower-left corner of pixel i is the
// lower-right corner of pixel i-1, and the top corners of row j are the bottom
// corners of row j-1 -- so each output pixel costs ONE 12-byte gather: the left
// corner arrives by DPP from the neighbouring lane and the top corners stay in
// registers.  Where the reference's wrap / clamp rules break that sharing
// (frame seam, first row of a run) the corner is loaded explicitly, so the
// result is the reference's for every pixel.
__device__ __forceinline__ uint32_t dpp_from_lane_below(uint32_t v) {
  // wave_shr:1 -- lane l receives lane l-1's value, lane 0 keeps its own
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint3 dpp_from_lane_below(const uint3 &v) {
  return make_uint3(dpp_from_lane_below(v.x), dpp_from_lane_below(v.y),
                    dpp_from_lane_below(v.z));
}

constexpr int kWalkCols = 63;
constexpr int kWalkBatch = 8;  // rows whose gathers are issued together

__global__ __launch_bounds__(256) void sample_rect_walk_kernel(
    uint8_t *__restrict__ dst, int out_w, int out_h, int out_stride_px,
    const uint32_t *__restrict__ sat, int src_w, int src_h,
    const int16_t *__restrict__ gx, const int16_t *__restrict__ gy, int cxp,
    int cyp, int rows) {
  const int lane = threadIdx.x & 63;
  const int c0 = __builtin_amdgcn_readfirstlane(
      ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * kWalkCols);
  if (c0 >= out_w) return;  // whole wave
  const int i = c0 - 1 + lane;
  const int ic = min(max(i, 0), out_w - 1);
  const AxisBox bx = sample_axis(cxp, gx[ic + 1], gx[ic], src_w, true);
  const bool writes = lane > 0 && i < out_w && bx.ok;
  // lane 0 never writes, so it never needs a left corner of its own
  const bool left_shared =
      lane == 0 || dpp_from_lane_below((uint32_t)bx.hi) == (uint32_t)bx.lo;
  const uint32_t dxw = (uint32_t)(bx.hi - bx.lo);

  const int j0 = blockIdx.y * rows, j1 = min(j0 + rows, out_h);
  int prev_hi = -1;
  uint3 p_br = make_uint3(0, 0, 0), p_bl = make_uint3(0, 0, 0);
  for (int jb = j0; jb < j1; jb += kWalkBatch) {
    // wave-uniform row boxes of the batch, then all of its gathers at once
    AxisBox by[kWalkBatch];
    bool any = false;
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      const int j = min(jb + r, out_h - 1);
      by[r] = sample_axis(cyp, gy[j + 1], gy[j], src_h, false);
      by[r].ok = by[r].ok && (jb + r < j1);
      any = any || by[r].ok;
    }
    if (!any) {
      prev_hi = -1;
      continue;
    }
    uint3 brs[kWalkBatch];
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r)  // clamped corners are always in range
      brs[r] = load_sat3(sat, (size_t)by[r].hi * src_w + bx.hi);
#pragma unroll
    for (int r = 0; r < kWalkBatch; ++r) {
      if (!by[r].ok) {
        prev_hi = -1;
        continue;
      }
      const int j = jb + r;
      const uint3 br = brs[r];
      uint3 bl = dpp_from_lane_below(br);
      if (!left_shared) bl = load_sat3(sat, (size_t)by[r].hi * src_w + bx.lo);
      uint3 tr, tl;
      if (by[r].lo == prev_hi) {
        tr = p_br;
        tl = p_bl;
      } else {
        tr = load_sat3(sat, (size_t)by[r].lo * src_w + bx.hi);
        tl = dpp_from_lane_below(tr);
        if (!left_shared) tl = load_sat3(sat, (size_t)by[r].lo * src_w + bx.lo);
      }
      if (writes) {
        const uint32_t area = dxw * (uint32_t)(by[r].hi - by[r].lo);
        uint8_t *o = dst + ((size_t)j * out_stride_px + i) * 4;
        const uint32_t rr = udiv_exact(br.x - tr.x + tl.x - bl.x, area);
        const uint32_t gg = udiv_exact(br.y - tr.y + tl.y - bl.y, area);
        const uint32_t bb = udiv_exact(br.z - tr.z + tl.z - bl.z, area);
        *reinterpret_cast<uint16_t *>(o) = (uint16_t)((rr & 0xffu) | ((gg & 0xffu) << 8));
        o[2] = (uint8_t)bb;
      }
      p_br = br;
      p_bl = bl;
      prev_hi = by[r].hi;
    }
  }
}

// ---------------------------------------------------------------------------
// decode_kernel (src/sat_decoder_decode_kernel.cl:1-58): 1x1 boxes.
__global__ __launch_bounds__(256) void decode_kernel(
    uint8_t *__restrict__ dst, int dst_linesize, int dst_bpp,
    const uint32_t *__restrict__ sat, int width, int height) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
  const size_t sl = (size_t)3 * width;
  uint8_t *o = dst + (size_t)y * dst_linesize + (size_t)x * dst_bpp;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint32_t v = sat[y * sl + 3 * x + c];
    if (x > 0 && y > 0)
      v = v - sat[(y - 1) * sl + 3 * x + c] + sat[(y - 1) * sl + 3 * (x - 1) + c] -
          sat[y * sl + 3 * (x - 1) + c];
    else if (x > 0)
      v -= sat[y * sl + 3 * (x - 1) + c];
    else if (y > 0)
      v -= sat[(y - 1) * sl + 3 * x + c];
    else
      v = sat[c];
    o[c] = (uint8_t)min(v, 255u);
  }
}

// ---------------------------------------------------------------------------
// interpolate_rect_kernel (src/sat_decoder_interpolate_kernel.cl:1-152).
// tx / ty hold, per pixel offset from the gaze centre, the inverse map u, the
// forward map of u and of its neighbour (host_tables.h: InterpAxisEntry).
__device__ __forceinline__ float mixf(float a, float b, float t) {
  return a + (b - a) * t;  // compiled with -ffp-contract=off: never fused
}

__global__ __launch_bounds__(256) void interpolate_rect_kernel(
    uint32_t *__restrict__ dst, int out_w, int out_h,
    const uint32_t *__restrict__ src, int src_w, int src_h,
    const int4 *__restrict__ tx, int range_x, const int4 *__restrict__ ty,
    int range_y, int cxp, int cyp) {
  const int x0 = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x0 >= out_w || y >= out_h) return;
  int x = x0;
  bool x_offset = false;
  if (x - cxp > out_w / 2) {  // :27-33
    x -= out_w;
    x_offset = true;
  } else if (x - cxp < (-out_w) / 2) {
    x += out_w;
    x_offset = true;
  }
  const int dx = x - cxp, dy = y - cyp;
  const int4 ex = tx[dx + range_x];  // {u, dcalc, dmin, du}
  const int4 ey = ty[dy + range_y];
  const int u = ex.x, v = ey.x;
  const int rw = src_w, rh = src_h;
  uint32_t out;
  if (ex.y == dx && ey.y == dy) {  // :67-72 exact hit -> copy
    const int r = min(max(v + rh / 2, 0), src_h - 1);
    const int c = min(max(u + rw / 2, 0), src_w - 1);
    out = src[(size_t)r * src_w + c] & 0x00ffffffu;
  } else {
    const int du = ex.w, dv = ey.w;
    const int a0 = cxp + ex.z, a1 = cxp + ex.y;
    const int b0 = cyp + ey.z, b1 = cyp + ey.y;
    const int min_x = min(a0, a1), max_x = max(a0, a1);
    const int min_y = min(b0, b1), max_y = max(b0, b1);
    int min_u = min(u, u + du), max_u = max(u, u + du);
    int min_v = min(v, v + dv), max_v = max(v, v + dv);
    if (min_x < 0 && !x_offset) min_u = max_u;  // :105-116
    if (max_x >= out_w && !x_offset) max_u = min_u;
    if (min_y < 0) min_v = max_v;
    if (max_y >= out_h) max_v = min_v;
    const int r0 = min(max(min_v + rh / 2, 0), src_h - 1);
    const int r1 = min(max(max_v + rh / 2, 0), src_h - 1);
    const int c0 = min(max(min_u + rw / 2, 0), src_w - 1);
    const int c1 = min(max(max_u + rw / 2, 0), src_w - 1);
    const uint32_t tl = src[(size_t)r0 * src_w + c0];
    const uint32_t tr = src[(size_t)r0 * src_w + c1];
    const uint32_t bl = src[(size_t)r1 * src_w + c0];
    const uint32_t br = src[(size_t)r1 * src_w + c1];
    const float yr = max_y == min_y
                         ? 0.0f
                         : fminf(fmaxf((float)(y - min_y) / (float)(max_y - min_y), 0.0f), 1.0f);
    const float xr = max_x == min_x
                         ? 0.0f
                         : fminf(fmaxf((float)(x - min_x) / (float)(max_x - min_x), 0.0f), 1.0f);
    out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float l = mixf((float)((tl >> (8 * c)) & 0xffu), (float)((bl >> (8 * c)) & 0xffu), yr);
      const float r = mixf((float)((tr >> (8 * c)) & 0xffu), (float)((br >> (8 * c)) & 0xffu), yr);
      out |= ((uint32_t)(int)mixf(l, r, xr) & 0xffu) << (8 * c);
    }
  }
  dst[(size_t)y * out_w + x0] = out;
}

int upload(f360_ctx *ctx, f360::DevBuf &buf, const void *host, size_t bytes) {
  // The tables may still be read by kernels queued earlier on the stream.
  F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
  int st = buf.reserve(bytes);
  if (st != F360_OK) return st;
  F360_HIP_TRY(hipMemcpy(buf.p, host, bytes, hipMemcpyHostToDevice));
  return F360_OK;
}

int ensure_interp_tables(f360_sat_decoder *dec, int w, int h, int rw, int rh,
                         int need_dx, int need_dy) {
  const bool same_geo =
      dec->it_w == w && dec->it_h == h && dec->it_rw == rw && dec->it_rh == rh;
  if (same_geo && dec->it_dx >= need_dx && dec->it_dy >= need_dy) return F360_OK;
  // Default coverage: every gaze centre in [-0.5, 1.5] x [-1, 2].
  const int rx = need_dx > w ? need_dx : w;
  const int ry = need_dy > 2 * h ? need_dy : 2 * h;
  std::vector<f360::InterpAxisEntry> t;
  f360::build_interp_axis(t, rx, w, rw);
  int st = upload(dec->ctx, dec->itx_dev, t.data(), t.size() * sizeof(t[0]));
  if (st != F360_OK) return st;
  f360::build_interp_axis(t, ry, h, rh);
  st = upload(dec->ctx, dec->ity_dev, t.data(), t.size() * sizeof(t[0]));
  if (st != F360_OK) return st;
  dec->it_w = w;
  dec->it_h = h;
  dec->it_rw = rw;
  dec->it_rh = rh;
  dec->it_dx = rx;
  dec->it_dy = ry;
  return F360_OK;
}

inline int iabs(int v) { return v < 0 ? -v : v; }
inline int imax(int a, int b) { return a > b ? a : b; }

}  // namespace

extern "C" {

int f360_satdec_create(f360_ctx *ctx, f360_sat_decoder **out) {
  F360_REQUIRE(ctx && out, "f360_satdec_create: null argument");
  f360_sat_decoder *d = new f360_sat_decoder();
  d->ctx = ctx;
  *out = d;
  return F360_OK;
}

int f360_satdec_destroy(f360_sat_decoder *dec) {
  if (!dec) return F360_OK;
  (void)hipStreamSynchronize(dec->ctx->stream);
  dec->gx_dev.release();
  dec->gy_dev.release();
  dec->itx_dev.release();
  dec->ity_dev.release();
  delete dec;
  return F360_OK;
}

int f360_satdec_initialize_grid(f360_sat_decoder *dec, int target_width,
                                int target_height, int source_width,
                                int source_height) {
  F360_REQUIRE(dec, "f360_satdec_initialize_grid: null decoder");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2,
               "f360_satdec_initialize_grid: bad geometry %dx%d <- %dx%d",
               target_width, target_height, source_width, source_height);
  if (dec->gw == target_width && dec->gh == target_height &&
      dec->sw == source_width && dec->sh == source_height && dec->gx_dev.p)
    return F360_OK;
  F360_HIP_TRY(hipSetDevice(dec->ctx->device));
  f360::build_satdec_grid_axis(dec->gx_host, target_width, source_width);
  f360::build_satdec_grid_axis(dec->gy_host, target_height, source_height);
  int st = upload(dec->ctx, dec->gx_dev, dec->gx_host.data(),
                  dec->gx_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  st = upload(dec->ctx, dec->gy_dev, dec->gy_host.data(),
              dec->gy_host.size() * sizeof(int16_t));
  if (st != F360_OK) return st;
  dec->gw = target_width;
  dec->gh = target_height;
  dec->sw = source_width;
  dec->sh = source_height;
  return F360_OK;
}

int f360_satdec_export_grid(f360_sat_decoder *dec, int16_t *grid_host) {
  F360_REQUIRE(dec && grid_host, "f360_satdec_export_grid: null argument");
  if (!dec->gx_dev.p) {
    f360::set_error("f360_satdec_export_grid: grid not initialised");
    return F360_ERR_NOT_INITIALIZED;
  }
  // read back what the kernels actually use
  std::vector<int16_t> gx(dec->gx_host.size()), gy(dec->gy_host.size());
  F360_HIP_TRY(hipStreamSynchronize(dec->ctx->stream));
  F360_HIP_TRY(hipMemcpy(gx.data(), dec->gx_dev.p, gx.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  F360_HIP_TRY(hipMemcpy(gy.data(), dec->gy_dev.p, gy.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost));
  const int gw = dec->gw + 1;
  for (int ty = 0; ty <= dec->gh; ++ty)
    for (int tx = 0; tx <= dec->gw; ++tx) {
      grid_host[((size_t)ty * gw + tx) * 2 + 0] = gx[(size_t)tx];
      grid_host[((size_t)ty * gw + tx) * 2 + 1] = gy[(size_t)ty];
    }
  return F360_OK;
}

int f360_satdec_sample_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                            int target_width, int target_height,
                            int target_linesize, const uint32_t *sat_dev,
                            int source_width, int source_height, float center_x,
                            float center_y) {
  F360_REQUIRE(dec, "f360_satdec_sample_rect: null decoder");
  F360_REQUIRE(target_dev && sat_dev, "f360_satdec_sample_rect: null buffer");
  F360_REQUIRE(target_width >= 1 && target_height >= 1 && source_width >= 2 &&
                   source_height >= 2 && target_linesize >= 4 * target_width,
               "f360_satdec_sample_rect: bad geometry");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_satdec_sample_rect: gaze centre out of range");
  if (!dec->gx_dev.p) {  // src/sat_decoder.cc:312-317
    int st = f360_satdec_initialize_grid(dec, target_width, target_height,
                                         source_width, source_height);
    if (st != F360_OK) return st;
  }
  F360_REQUIRE(dec->gw == target_width && dec->gh == target_height,
               "f360_satdec_sample_rect: grid was initialised for %dx%d, not %dx%d",
               dec->gw, dec->gh, target_width, target_height);
  // (int2)(center.x * source_width, center.y * source_height): float multiply,
  // truncation (:176-179)
  const int cxp = (int)(center_x * (float)source_width);
  const int cyp = (int)(center_y * (float)source_height);
  f360_ctx *ctx = dec->ctx;
  const dim3 grid((target_width + 63) / 64, (target_height + 3) / 4);
  if (ctx->opt_sample_variant == 1) {
    const int rows = ctx->opt_walk_rows;
    const dim3 wgrid((target_width + 4 * kWalkCols - 1) / (4 * kWalkCols),
                     (target_height + rows - 1) / rows);
    f360::KernelSpan span(ctx, f360::kSampleRect, f360::take_profile_slot(ctx));
    hipLaunchKernelGGL(sample_rect_walk_kernel, wgrid, dim3(256), 0, ctx->stream,
                       target_dev, target_width, target_height,
                       target_linesize / 4, sat_dev, source_width, source_height,
                       dec->gx_dev.as<int16_t>(), dec->gy_dev.as<int16_t>(), cxp,
                       cyp, rows);
  } else {
    f360::KernelSpan span(ctx, f360::kSampleRect, f360::take_profile_slot(ctx));
    hipLaunchKernelGGL(sample_rect_kernel, grid, dim3(256), 0, ctx->stream,
                       target_dev, target_width, target_height,
                       target_linesize / 4, sat_dev, source_width, source_height,
                       dec->gx_dev.as<int16_t>(), dec->gy_dev.as<int16_t>(), cxp,
                       cyp);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_satdec_decode(f360_sat_decoder *dec, uint8_t *target_dev,
                       int target_linesize, const uint32_t *sat_dev, int width,
                       int height) {
  F360_REQUIRE(dec, "f360_satdec_decode: null decoder");
  F360_REQUIRE(target_dev && sat_dev, "f360_satdec_decode: null buffer");
  F360_REQUIRE(width >= 1 && height >= 1 && target_linesize / width >= 3,
               "f360_satdec_decode: bad geometry");
  const dim3 grid((width + 63) / 64, (height + 3) / 4);
  hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, dec->ctx->stream,
                     target_dev, target_linesize, target_linesize / width, sat_dev,
                     width, height);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

int f360_satdec_interpolate_rect(f360_sat_decoder *dec, uint8_t *target_dev,
                                 int target_width, int target_height,
                                 int target_linesize, const uint8_t *source_dev,
                                 int source_width, int source_height,
                                 int source_linesize, float center_x,
                                 float center_y) {
  (void)target_linesize;  // the reference kernel never receives the linesizes
  (void)source_linesize;
  F360_REQUIRE(dec, "f360_satdec_interpolate_rect: null decoder");
  F360_REQUIRE(target_dev && source_dev, "f360_satdec_interpolate_rect: null buffer");
  F360_REQUIRE(target_width >= 2 && target_height >= 2 && source_width >= 1 &&
                   source_height >= 1,
               "f360_satdec_interpolate_rect: bad geometry");
  F360_REQUIRE(((uintptr_t)target_dev % 4) == 0 && ((uintptr_t)source_dev % 4) == 0,
               "f360_satdec_interpolate_rect: buffers must be 4-byte aligned");
  F360_REQUIRE(std::fabs(center_x) <= 16.0f && std::fabs(center_y) <= 16.0f,
               "f360_satdec_interpolate_rect: gaze centre out of range");
  const int cxp = (int)(center_x * (float)target_width);
  const int cyp = (int)(center_y * (float)target_height);
  // largest |offset| any pixel can have after the single +-W wrap
  const int need_dx = imax(iabs(0 - cxp), iabs(target_width - 1 - cxp)) + 1;
  const int need_dy = imax(iabs(0 - cyp), iabs(target_height - 1 - cyp)) + 1;
  F360_HIP_TRY(hipSetDevice(dec->ctx->device));
  int st = ensure_interp_tables(dec, target_width, target_height, source_width,
                                source_height, need_dx, need_dy);
  if (st != F360_OK) return st;
  const dim3 grid((target_width + 63) / 64, (target_height + 3) / 4);
  f360::KernelSpan span(dec->ctx, f360::kInterpolateRect,
                        f360::take_profile_slot(dec->ctx));
  hipLaunchKernelGGL(interpolate_rect_kernel, grid, dim3(256), 0, dec->ctx->stream,
                     reinterpret_cast<uint32_t *>(target_dev), target_width,
                     target_height, reinterpret_cast<const uint32_t *>(source_dev),
                     source_width, source_height, dec->itx_dev.as<int4>(),
                     dec->it_dx, dec->ity_dev.as<int4>(), dec->it_dy, cxp, cyp);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

}  // extern "C"
