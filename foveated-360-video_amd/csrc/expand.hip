// expand.hip -- the "expand" debug views: every pixel of a reduced frame scattered back to the
// place of the full frame it was sampled from (nearest "un-warp" without interpolation).
//
// The reference has them on the CPU only: SATDecoder::ExpandSampledFrameRectCPU
// (src/sat_decoder.cc:555-616, copied as ImageSampler::ExpandSampledFrameRectCPU,
// src/image_sampler.cc:358-419) and ImageSampler::ExpandSampledFrameLogPolarCPU
// (src/image_sampler.cc:623-666).  Both maps are separable up to one float operation, so the
// transcendental part is two host tables per geometry (host_tables.cpp) and the kernels do the
// reference's float position arithmetic:  x_pos = (int)(center_x * target_width + delta_x).
// Target pixels nothing lands on keep their value, as in the reference.
#include "f360_internal.h"
#include "host_tables.h"

#include <cmath>
#include <cstring>

namespace {

struct ExpandArgs {
  uint8_t *dst;
  const uint8_t *src;
  int dst_w, dst_h, dst_linesize, dst_bpp;
  int src_w, src_h, src_linesize, src_bpp;
  float cxw, cyh;  // center_x * target_width, center_y * target_height (float products)
  // rect: integer offsets per reduced column / row
  const int32_t *dx, *dy;
  // log-polar: float radius per column, double cos / sin per row
  const float *radius;
  const double *cs, *sn;
  uint32_t *keys;  // log-polar: per target pixel, 1 + index of the last writer in loop order
};

__device__ __forceinline__ void copy_px(const ExpandArgs &a, int x_pos, int y_pos, int i, int j) {
  uint8_t *t = a.dst + (size_t)y_pos * a.dst_linesize + (size_t)x_pos * a.dst_bpp;
  const uint8_t *s = a.src + (size_t)j * a.src_linesize + (size_t)i * a.src_bpp;
  t[0] = s[0];
  t[1] = s[1];
  t[2] = s[2];
}

// The offsets grow strictly with the column / row (|u| grows by one per step and the exponential
// term never shrinks), so positions are non-decreasing and two source columns can share a target
// column only as neighbours -- which happens where the float sum crosses zero, because the
// conversion truncates towards zero ((int)-0.6 == (int)0.4 == 0).  The reference's loops let
// the later column / row win, so a pixel writes iff it is the last of its column and of its row
// to land there.
__global__ __launch_bounds__(256) void expand_rect_kernel(const ExpandArgs a) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= a.src_w || j >= a.src_h) return;
  const int x_pos = (int)(a.cxw + (float)a.dx[i]);
  const int y_pos = (int)(a.cyh + (float)a.dy[j]);
  if (i + 1 < a.src_w && (int)(a.cxw + (float)a.dx[i + 1]) == x_pos) return;
  if (j + 1 < a.src_h && (int)(a.cyh + (float)a.dy[j + 1]) == y_pos) return;
  if (x_pos >= 0 && x_pos < a.dst_w && y_pos >= 0 && y_pos < a.dst_h) copy_px(a, x_pos, y_pos, i, j);
}

// Log-polar: near the centre many source pixels land on one target pixel and the reference's
// loops (i outer, j inner) let the last one win.  Pass 1 records per target pixel the largest
// i * src_h + j (+1) that lands there, pass 2 lets exactly that pixel write.
__device__ __forceinline__ bool logpolar_target(const ExpandArgs &a, int i, int j, int &x_pos,
                                                int &y_pos) {
  const double r = (double)a.radius[i];
  const float delta_x = (float)(r * a.cs[j]);
  const float delta_y = (float)(r * a.sn[j]);
  x_pos = (int)(a.cxw + delta_x);
  y_pos = (int)(a.cyh + delta_y);
  return x_pos >= 0 && x_pos < a.dst_w && y_pos >= 0 && y_pos < a.dst_h;
}

template <int PASS>
__global__ __launch_bounds__(256) void expand_logpolar_kernel(const ExpandArgs a) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= a.src_w || j >= a.src_h) return;
  int x_pos, y_pos;
  if (!logpolar_target(a, i, j, x_pos, y_pos)) return;
  const uint32_t order = (uint32_t)i * (uint32_t)a.src_h + (uint32_t)j + 1u;
  uint32_t *key = a.keys + (size_t)y_pos * a.dst_w + x_pos;
  if (PASS == 0)
    atomicMax(key, order);
  else if (*key == order)
    copy_px(a, x_pos, y_pos, i, j);
}

int expand_common(f360_ctx *ctx, ExpandArgs &a, uint8_t *dst, int dst_w, int dst_h,
                  int dst_linesize, const uint8_t *src, int src_w, int src_h, int src_linesize,
                  float cx, float cy, const char *who) {
  F360_REQUIRE(ctx, "%s: null context", who);
  F360_REQUIRE(dst && src, "%s: null buffer", who);
  F360_REQUIRE(dst_w >= 1 && dst_h >= 1 && src_w >= 1 && src_h >= 1, "%s: bad geometry", who);
  F360_REQUIRE(f360::dims_ok({dst_w, dst_h, src_w, src_h}), "f360_expand: a dimension exceeds 65536");
  F360_REQUIRE(dst_linesize / dst_w >= 3 && src_linesize / src_w >= 3,
               "%s: linesize gives fewer than 3 bytes per pixel", who);
  F360_REQUIRE(std::fabs(cx) <= 16.0f && std::fabs(cy) <= 16.0f, "%s: gaze centre out of range",
               who);
  a.dst = dst;
  a.src = src;
  a.dst_w = dst_w;
  a.dst_h = dst_h;
  a.dst_linesize = dst_linesize;
  a.dst_bpp = dst_linesize / dst_w;  // :569
  a.src_w = src_w;
  a.src_h = src_h;
  a.src_linesize = src_linesize;
  a.src_bpp = src_linesize / src_w;  // :562
  a.cxw = cx * (float)dst_w;
  a.cyh = cy * (float)dst_h;
  return F360_OK;
}

}  // namespace

extern "C" int f360_expand_rect(f360_ctx *ctx, uint8_t *dst_dev, int dst_w, int dst_h,
                                int dst_linesize, const uint8_t *src_dev, int src_w, int src_h,
                                int src_linesize, float center_x, float center_y) {
  ExpandArgs a{};
  int st = expand_common(ctx, a, dst_dev, dst_w, dst_h, dst_linesize, src_dev, src_w, src_h,
                         src_linesize, center_x, center_y, "f360_expand_rect");
  if (st != F360_OK) return st;
  F360_BIND_DEVICE(ctx);
  if (!(ctx->ex_kind == 0 && ctx->ex_w == src_w && ctx->ex_h == src_h && ctx->ex_tw == dst_w &&
        ctx->ex_th == dst_h)) {
    std::vector<int32_t> dx, dy;
    f360::build_expand_axis(dx, src_w, dst_w);
    f360::build_expand_axis(dy, src_h, dst_h);
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));  // earlier calls may still read the old tables
    st = ctx->ex_tables.reserve((dx.size() + dy.size()) * sizeof(int32_t));
    if (st != F360_OK) return st;
    F360_HIP_TRY(hipMemcpy(ctx->ex_tables.p, dx.data(), dx.size() * 4, hipMemcpyHostToDevice));
    F360_HIP_TRY(hipMemcpy(ctx->ex_tables.as<int32_t>() + dx.size(), dy.data(), dy.size() * 4,
                           hipMemcpyHostToDevice));
    ctx->ex_kind = 0;
    ctx->ex_w = src_w;
    ctx->ex_h = src_h;
    ctx->ex_tw = dst_w;
    ctx->ex_th = dst_h;
  }
  a.dx = ctx->ex_tables.as<int32_t>();
  a.dy = a.dx + src_w;
  const bool prof = f360::take_profile_slot(ctx);
  f360::KernelSpan span(ctx, f360::kExpand, prof);
  hipLaunchKernelGGL(expand_rect_kernel, dim3((src_w + 63) / 64, (src_h + 3) / 4), dim3(256), 0,
                     ctx->stream, a);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}

extern "C" int f360_expand_logpolar(f360_ctx *ctx, uint8_t *dst_dev, int dst_w, int dst_h,
                                    int dst_linesize, const uint8_t *src_dev, int src_w,
                                    int src_h, int src_linesize, float center_x,
                                    float center_y) {
  ExpandArgs a{};
  int st = expand_common(ctx, a, dst_dev, dst_w, dst_h, dst_linesize, src_dev, src_w, src_h,
                         src_linesize, center_x, center_y, "f360_expand_logpolar");
  if (st != F360_OK) return st;
  F360_REQUIRE((uint64_t)src_w * (uint64_t)src_h < 0xffffffffull,
               "f360_expand_logpolar: source too large");
  F360_BIND_DEVICE(ctx);
  const size_t n_keys = (size_t)dst_w * dst_h;
  const size_t table_bytes = (size_t)src_h * 2 * sizeof(double) + (size_t)src_w * sizeof(float);
  if (!(ctx->ex_kind == 1 && ctx->ex_w == src_w && ctx->ex_h == src_h && ctx->ex_tw == dst_w &&
        ctx->ex_th == dst_h)) {
    std::vector<float> radius;
    std::vector<double> cs, sn;
    f360::build_expand_logpolar_axes(radius, cs, sn, src_w, src_h);
    F360_HIP_TRY(hipStreamSynchronize(ctx->stream));
    st = ctx->ex_tables.reserve(table_bytes);
    if (st != F360_OK) return st;
    st = ctx->ex_keys.reserve(n_keys * sizeof(uint32_t));
    if (st != F360_OK) return st;
    char *base = ctx->ex_tables.as<char>();
    F360_HIP_TRY(hipMemcpy(base, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
    F360_HIP_TRY(hipMemcpy(base + cs.size() * 8, sn.data(), sn.size() * 8, hipMemcpyHostToDevice));
    F360_HIP_TRY(hipMemcpy(base + (cs.size() + sn.size()) * 8, radius.data(), radius.size() * 4,
                           hipMemcpyHostToDevice));
    ctx->ex_kind = 1;
    ctx->ex_w = src_w;
    ctx->ex_h = src_h;
    ctx->ex_tw = dst_w;
    ctx->ex_th = dst_h;
  }
  a.cs = ctx->ex_tables.as<double>();
  a.sn = a.cs + src_h;
  a.radius = reinterpret_cast<const float *>(a.sn + src_h);
  a.keys = ctx->ex_keys.as<uint32_t>();
  F360_HIP_TRY(hipMemsetAsync(a.keys, 0, n_keys * sizeof(uint32_t), ctx->stream));
  const bool prof = f360::take_profile_slot(ctx);
  f360::KernelSpan span(ctx, f360::kExpand, prof);
  const dim3 grid((src_w + 63) / 64, (src_h + 3) / 4);
  hipLaunchKernelGGL(expand_logpolar_kernel<0>, grid, dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL(expand_logpolar_kernel<1>, grid, dim3(256), 0, ctx->stream, a);
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
