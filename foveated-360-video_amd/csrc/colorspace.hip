// colorspace.hip -- planar YUV 4:2:0 -> RGB0 on the device.
//
// Replaces the CPU sws_scale call in front of the hot path: VideoDecoder::GetFrame converts every
// decoded frame with sws_getContext(w, h, yuv420p, w, h, AV_PIX_FMT_RGB0, SWS_BILINEAR) +
// sws_scale (src/video_decoder.cc:167-170,222-224) and the server then uploads the 4-byte
// pixels (src/video_server.cc:291-300).  With this kernel the three planes are uploaded instead
// (1.5 instead of 4 bytes per pixel over PCIe); f360_sat_encode_yuv420p (sat_encode.hip) goes
// one step further and never writes the RGB0 frame at all.  Arithmetic: yuv_device.h.
#include "f360_internal.h"
#include "host_tables.h"

namespace {

using f360::YuvConsts;
using f360::YuvPlanes;

typedef uint32_t u32x4_c __attribute__((ext_vector_type(4)));

// A lane converts a 4 x 2 block (two chroma pairs): two luma dwords, one 16-bit load per
// chroma plane, two 16-byte stores.
template <int MODEL>
__global__ __launch_bounds__(256) void yuv420p_to_rgb0_kernel(uint8_t *__restrict__ dst,
                                                              int dst_linesize, const YuvPlanes p,
                                                              int width, int height,
                                                              const YuvConsts k) {
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
  if (x0 >= width || y0 >= height) return;
  const uint32_t ya = *reinterpret_cast<const uint32_t *>(p.y + (size_t)y0 * p.y_linesize + x0);
  const uint32_t yb =
      *reinterpret_cast<const uint32_t *>(p.y + (size_t)(y0 + 1) * p.y_linesize + x0);
  const uint32_t u =
      *reinterpret_cast<const uint16_t *>(p.u + (size_t)(y0 >> 1) * p.u_linesize + (x0 >> 1));
  const uint32_t v =
      *reinterpret_cast<const uint16_t *>(p.v + (size_t)(y0 >> 1) * p.v_linesize + (x0 >> 1));
  const uint32_t uv = u | (v << 16);
  uint32_t a[4], b[4];
  f360::yuv_pixels4<MODEL>(k, ya, uv, a);
  f360::yuv_pixels4<MODEL>(k, yb, uv, b);
  const uint32_t alpha = 0xff000000u;  // yuv2rgb.c:983-984 / SET_EMPTY_ALPHA
#ifdef F360_NO_NT_STORES
  *reinterpret_cast<u32x4_c *>(dst + (size_t)y0 * dst_linesize + (size_t)x0 * 4) =
      u32x4_c{a[0] | alpha, a[1] | alpha, a[2] | alpha, a[3] | alpha};
  *reinterpret_cast<u32x4_c *>(dst + (size_t)(y0 + 1) * dst_linesize + (size_t)x0 * 4) =
      u32x4_c{b[0] | alpha, b[1] | alpha, b[2] | alpha, b[3] | alpha};
#else  // streaming output: non-temporal, see sat_encode.hip
  __builtin_nontemporal_store(
      (u32x4_c{a[0] | alpha, a[1] | alpha, a[2] | alpha, a[3] | alpha}),
      reinterpret_cast<u32x4_c *>(dst + (size_t)y0 * dst_linesize + (size_t)x0 * 4));
  __builtin_nontemporal_store(
      (u32x4_c{b[0] | alpha, b[1] | alpha, b[2] | alpha, b[3] | alpha}),
      reinterpret_cast<u32x4_c *>(dst + (size_t)(y0 + 1) * dst_linesize + (size_t)x0 * 4));
#endif
}

// Any width, any alignment: one pixel per thread, byte loads.
template <int MODEL>
__global__ __launch_bounds__(256) void yuv420p_to_rgb0_px_kernel(uint8_t *__restrict__ dst,
                                                                 int dst_linesize,
                                                                 const YuvPlanes p, int width,
                                                                 int height, const YuvConsts k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
  const int Y = p.y[(size_t)y * p.y_linesize + x];
  const int U = p.u[(size_t)(y >> 1) * p.u_linesize + (x >> 1)];
  const int V = p.v[(size_t)(y >> 1) * p.v_linesize + (x >> 1)];
  const uint32_t px = f360::yuv_pixel<MODEL>(k, Y, f360::chroma_terms<MODEL>(k, U, V));
  uint8_t *o = dst + (size_t)y * dst_linesize + (size_t)x * 4;
  o[0] = (uint8_t)px;
  o[1] = (uint8_t)(px >> 8);
  o[2] = (uint8_t)(px >> 16);
  o[3] = 255;
}

}  // namespace

extern "C" int f360_yuv420p_to_rgb0(f360_ctx *ctx, uint8_t *dst_dev, int dst_linesize,
                                    const uint8_t *y_dev, const uint8_t *u_dev,
                                    const uint8_t *v_dev, int y_linesize, int u_linesize,
                                    int v_linesize, int width, int height) {
  F360_REQUIRE(ctx, "f360_yuv420p_to_rgb0: null context");
  F360_REQUIRE(dst_dev && y_dev && u_dev && v_dev, "f360_yuv420p_to_rgb0: null buffer");
  // an odd height sends sws_scale through its generic scaler (swscale_unscaled.c:1933), whose
  // arithmetic is different: not provided
  F360_REQUIRE(width >= 1 && height >= 2 && height % 2 == 0,
               "f360_yuv420p_to_rgb0: bad size %dx%d (height must be even)", width, height);
  F360_REQUIRE(dst_linesize >= 4 * width && y_linesize >= width &&
                   u_linesize >= (width + 1) / 2 && v_linesize >= (width + 1) / 2,
               "f360_yuv420p_to_rgb0: linesize too small");
  F360_BIND_DEVICE(ctx);
  YuvConsts k;
  f360::build_yuv2rgb_consts(k);
  const YuvPlanes p{y_dev, u_dev, v_dev, y_linesize, u_linesize, v_linesize};
  const bool vec = width % 4 == 0 && dst_linesize % 16 == 0 && y_linesize % 4 == 0 &&
                   u_linesize % 2 == 0 && v_linesize % 2 == 0 &&
                   (reinterpret_cast<uintptr_t>(dst_dev) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y_dev) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(u_dev) & 1) == 0 &&
                   (reinterpret_cast<uintptr_t>(v_dev) & 1) == 0;
  const bool prof = f360::take_profile_slot(ctx);
  f360::KernelSpan span(ctx, f360::kYuvToRgb, prof);
  const bool x86 = ctx->opt_yuv_model == 1;
  if (vec) {
    const dim3 grid((width / 4 + 63) / 64, (height / 2 + 3) / 4);
    if (x86)
      hipLaunchKernelGGL(yuv420p_to_rgb0_kernel<1>, grid, dim3(256), 0, ctx->stream, dst_dev,
                         dst_linesize, p, width, height, k);
    else
      hipLaunchKernelGGL(yuv420p_to_rgb0_kernel<0>, grid, dim3(256), 0, ctx->stream, dst_dev,
                         dst_linesize, p, width, height, k);
  } else {
    const dim3 grid((width + 63) / 64, (height + 3) / 4);
    if (x86)
      hipLaunchKernelGGL(yuv420p_to_rgb0_px_kernel<1>, grid, dim3(256), 0, ctx->stream, dst_dev,
                         dst_linesize, p, width, height, k);
    else
      hipLaunchKernelGGL(yuv420p_to_rgb0_px_kernel<0>, grid, dim3(256), 0, ctx->stream, dst_dev,
                         dst_linesize, p, width, height, k);
  }
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
