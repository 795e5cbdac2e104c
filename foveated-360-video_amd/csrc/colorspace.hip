// colorspace.hip -- planar YUV 4:2:0 <-> RGB0 on the device, the colour-space steps either side
// of the hot path (the second half of this file is the output side, RGB0 -> yuv420p).
//
// Replaces the CPU sws_scale call in front of the hot path: VideoDecoder::GetFrame converts every
// decoded frame with sws_getContext(w, h, yuv420p, w, h, AV_PIX_FMT_RGB0, SWS_BILINEAR) +
// sws_scale (src/video_decoder.cc:167-170,222-224) and the server then uploads the 4-byte
// pixels (src/video_server.cc:291-300).  With this kernel the three planes are uploaded instead
// (1.5 instead of 4 bytes per pixel over PCIe); f360_sat_encode_yuv420p (sat_three.hip) goes
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
#else  // streaming output: non-temporal, see sat_common.h
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


// ================================================================================================
// RGB0 -> planar YUV 4:2:0: the sws_scale of VideoEncoder::EncodeFrame in front of NVENC
// (src/video_encoder.cc:380-395: sws_getContext(w, h, RGB0, w, h, AV_PIX_FMT_YUV420P,
// SWS_BILINEAR) + sws_scale on the CPU).  On the device the reduced frame can leave as 1.5
// instead of 4 bytes per pixel.
//
// FFmpeg 4.2 has no special converter for this pair: the generic scaler runs (paths relative to
// include/FFmpeg42/libswscale/).  What it computes, for even sizes:
//   Y14 = (RY R + GY G + BY B + (32 << 14) + (1 << 8)) >> 9      input.c:252-275 (rgb32ToY_c; the
//         x86 asm ff_rgbaToY_sse2 has the same rounding constant, x86/input.asm)
//   U14 = (RU sR + GU sG + BU sB + (256 << 15) + (1 << 9)) >> 10  over the PAIR of pixels
//         (sums sR = R0 + R1 ...), V14 likewise                   input.c:304-346 (rgb32ToUV_half_c)
//   one-tap horizontal filter 1 << 14, shifted by 13: x15 = min(2 x14, 32767)   swscale.c:95-121
//   Y  = clip8((Y15 + 64) >> 7)                                   output.c:395-403
//   chroma rows are halved by the bilinear 2:1 filter initFilter produces (utils.c:331-726):
//   taps 512, 1536, 1536, 512 on source rows 2c - 1 .. 2c + 2, folded at the borders into
//   2048, 1536, 512 (first row) and 512, 1536, 2048 (last row);
//   C builds:   U = clip8(((64 << 12) + sum_j U15_j f_j) >> 19)    output.c:380-393
//   x86 builds: U = clip8((5 + sum_j ((U15_j f_j) >> 16)) >> 3) -- pmulhw truncates every tap,
//               5 = (64 + ((4 - 1) << 3)) >> 4 (x86/swscale.c:209-248) -- for every chroma row
//               but the last, which the C function computes (swscale.c:497-504).
// "yuv.model" selects the build reproduced, as on the input side.  The coefficients are the
// ITU-R 601 limited-range set of utils.c:811-821.
namespace {

struct Rgb2YuvConsts {
  int ry, gy, by, ru, gu, bu, rv, gv, bv;
};

__host__ __device__ inline Rgb2YuvConsts rgb2yuv_consts() {
  Rgb2YuvConsts k;
  k.ry = (int)(0.299 * 219 / 255 * (1 << 15) + 0.5);
  k.gy = (int)(0.587 * 219 / 255 * (1 << 15) + 0.5);
  k.by = (int)(0.114 * 219 / 255 * (1 << 15) + 0.5);
  k.ru = -(int)(0.169 * 224 / 255 * (1 << 15) + 0.5);
  k.gu = -(int)(0.331 * 224 / 255 * (1 << 15) + 0.5);
  k.bu = (int)(0.500 * 224 / 255 * (1 << 15) + 0.5);
  k.rv = (int)(0.500 * 224 / 255 * (1 << 15) + 0.5);
  k.gv = -(int)(0.419 * 224 / 255 * (1 << 15) + 0.5);
  k.bv = -(int)(0.081 * 224 / 255 * (1 << 15) + 0.5);
  return k;
}

__device__ __forceinline__ int luma15(const Rgb2YuvConsts &k, uint32_t px) {
  const int r = px & 0xff, g = (px >> 8) & 0xff, b = (px >> 16) & 0xff;
  const int y14 = (k.ry * r + k.gy * g + k.by * b + (32 << 14) + (1 << 8)) >> 9;
  return min(2 * y14, 32767);
}
// chroma of a pixel pair, 15 bits: .x = U, .y = V
__device__ __forceinline__ int2 chroma15(const Rgb2YuvConsts &k, uint32_t p0, uint32_t p1) {
  const int r = (int)(p0 & 0xff) + (int)(p1 & 0xff);
  const int g = (int)((p0 >> 8) & 0xff) + (int)((p1 >> 8) & 0xff);
  const int b = (int)((p0 >> 16) & 0xff) + (int)((p1 >> 16) & 0xff);
  const int u14 = (k.ru * r + k.gu * g + k.bu * b + (256 << 15) + (1 << 9)) >> 10;
  const int v14 = (k.rv * r + k.gv * g + k.bv * b + (256 << 15) + (1 << 9)) >> 10;
  return make_int2(min(2 * u14, 32767), min(2 * v14, 32767));
}
// clamp(v >> N, 0, 255), written as clamp-then-shift: hipcc 7.2 folds two adjacent
// clamp(x >> n, 0, 255) into v_ashr_pk_u8_i32 and then ORs further bytes into that register as
// if its upper half were zero, which on gfx950 it is not (the parity test found it, as it did in
// yuv_device.h in round 1); this form does not match the pattern
template <int N>
__device__ __forceinline__ uint32_t shift_clip8(int v) {
  return (uint32_t)(min(max(v, 0), (256 << N) - 1) >> N);
}

// A thread owns PAIRS chroma samples of one chroma row: 2 * PAIRS pixels of the four source
// rows its vertical filter needs, the luma of the two rows it is centred on.  PAIRS = 4: 32-byte
// row pieces, 8-byte luma stores, 4-byte chroma stores; PAIRS = 1: any even width / alignment.
template <int MODEL, int PAIRS>
__global__ __launch_bounds__(256) void rgb0_to_yuv420p_kernel(
    uint8_t *__restrict__ y_dst, uint8_t *__restrict__ u_dst, uint8_t *__restrict__ v_dst,
    int y_linesize, int u_linesize, int v_linesize, const uint8_t *__restrict__ src,
    int src_linesize, int width, int height, const Rgb2YuvConsts k) {
  const int c0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * PAIRS;  // first chroma column
  const int cy = blockIdx.y * 4 + (threadIdx.x >> 6);             // chroma row
  const int cw = width >> 1, ch = height >> 1;
  if (c0 >= cw || cy >= ch) return;
  // taps and source rows: interior 512 1536 1536 512 on rows 2cy-1 .. 2cy+2; the first row
  // folds the tap above the frame into its neighbour, the last row the tap below
  int rows[4], taps[4];
  if (cy == 0) {
    rows[0] = 0; rows[1] = 1; rows[2] = 2; rows[3] = 2;
    taps[0] = 2048; taps[1] = 1536; taps[2] = 512; taps[3] = 0;
  } else if (cy == ch - 1) {
    rows[0] = height - 3; rows[1] = height - 3; rows[2] = height - 2; rows[3] = height - 1;
    taps[0] = 0; taps[1] = 512; taps[2] = 1536; taps[3] = 2048;
  } else {
    for (int j = 0; j < 4; ++j) {
      rows[j] = 2 * cy - 1 + j;
      taps[j] = j == 0 || j == 3 ? 512 : 1536;
    }
  }
  // x86 builds: every chroma row but the last through the truncating 16-bit multiply
  const bool mmx = MODEL == 1 && cy < ch - 1;
  int acc_u[PAIRS], acc_v[PAIRS];
#pragma unroll
  for (int q = 0; q < PAIRS; ++q) acc_u[q] = acc_v[q] = mmx ? 5 : (64 << 12);
  uint32_t ylo0[(PAIRS + 1) / 2], ylo1[(PAIRS + 1) / 2];  // luma bytes of rows 2cy, 2cy+1
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint8_t *row = src + (size_t)rows[j] * src_linesize + (size_t)c0 * 8;
    uint32_t px[2 * PAIRS];
    if (PAIRS == 4) {
      typedef uint32_t u32x4_r __attribute__((ext_vector_type(4)));
      const u32x4_r a = *reinterpret_cast<const u32x4_r *>(row);
      const u32x4_r b = *reinterpret_cast<const u32x4_r *>(row + 16);
      px[0] = a[0]; px[1] = a[1]; px[2] = a[2]; px[3] = a[3];
      px[4] = b[0]; px[5] = b[1]; px[6] = b[2]; px[7] = b[3];
    } else {
#pragma unroll
      for (int q = 0; q < 2 * PAIRS; ++q)
        px[q] = (uint32_t)row[4 * q] | ((uint32_t)row[4 * q + 1] << 8) | ((uint32_t)row[4 * q + 2] << 16);
    }
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
      const int2 c = chroma15(k, px[2 * q], px[2 * q + 1]);
      if (mmx) {
        acc_u[q] = (int)(int16_t)(acc_u[q] + ((c.x * taps[j]) >> 16));
        acc_v[q] = (int)(int16_t)(acc_v[q] + ((c.y * taps[j]) >> 16));
      } else {
        acc_u[q] += c.x * taps[j];
        acc_v[q] += c.y * taps[j];
      }
    }
    // the luma rows: source rows 2cy and 2cy+1 are taps 1 and 2 of an interior chroma row, taps
    // 0 and 1 of the first, 2 and 3 of the last
    const int lr = rows[j] - 2 * cy;
    if (lr == 0 || lr == 1) {
      uint32_t w[(PAIRS + 1) / 2];
#pragma unroll
      for (int q = 0; q < 2 * PAIRS; ++q) {
        const uint32_t yv = shift_clip8<7>(luma15(k, px[q]) + 64);
        if ((q & 3) == 0) w[q >> 2] = yv;
        else w[q >> 2] |= yv << (8 * (q & 3));
      }
#pragma unroll
      for (int q = 0; q < (PAIRS + 1) / 2; ++q) {
        if (lr == 0) ylo0[q] = w[q];
        else ylo1[q] = w[q];
      }
    }
  }
  uint32_t ub = 0, vb = 0;
#pragma unroll
  for (int q = 0; q < PAIRS; ++q) {
    ub |= (mmx ? shift_clip8<3>(acc_u[q]) : shift_clip8<19>(acc_u[q])) << (8 * q);
    vb |= (mmx ? shift_clip8<3>(acc_v[q]) : shift_clip8<19>(acc_v[q])) << (8 * q);
  }
  uint8_t *yo0 = y_dst + (size_t)(2 * cy) * y_linesize + (size_t)c0 * 2;
  uint8_t *yo1 = yo0 + y_linesize;
  uint8_t *uo = u_dst + (size_t)cy * u_linesize + c0, *vo = v_dst + (size_t)cy * v_linesize + c0;
  if (PAIRS == 4) {
    *reinterpret_cast<uint2 *>(yo0) = make_uint2(ylo0[0], ylo0[PAIRS == 4 ? 1 : 0]);
    *reinterpret_cast<uint2 *>(yo1) = make_uint2(ylo1[0], ylo1[PAIRS == 4 ? 1 : 0]);
    *reinterpret_cast<uint32_t *>(uo) = ub;
    *reinterpret_cast<uint32_t *>(vo) = vb;
  } else {
    yo0[0] = (uint8_t)ylo0[0];
    yo0[1] = (uint8_t)(ylo0[0] >> 8);
    yo1[0] = (uint8_t)ylo1[0];
    yo1[1] = (uint8_t)(ylo1[0] >> 8);
    uo[0] = (uint8_t)ub;
    vo[0] = (uint8_t)vb;
  }
}

// The same conversion for whole frames, walking down ("yuv.r2y_rows" chroma rows per wave): the
// four source rows of a chroma sample overlap the next sample's by two, so the kernel above reads
// every source row twice and evaluates its chroma twice.  Here a lane owns four chroma columns
// (eight pixels, 32 bytes of a source row) and keeps the 15-bit chroma of the window's rows
// 2c - 1 .. 2c + 2 in registers: per chroma row it loads two new source rows -- requested one
// chroma row ahead --, evaluates their luma and chroma once, and slides the window by two.
// Stores are issued from inline asm (non-temporal, invisible to the compiler's vmcnt
// bookkeeping: a store in the loop would otherwise turn every wait for a prefetched row into a
// full drain).  The borders use the window as it is: row
// -1 and row `height` are loaded clamped and carry tap 0, the other taps are the folded ones.
__device__ __forceinline__ void r2y_store8(uint8_t *base, uint32_t off, uint32_t a, uint32_t b) {
  typedef uint32_t u32x2_r __attribute__((ext_vector_type(2)));
  const u32x2_r v{a, b};
  asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void r2y_store4(uint8_t *base, uint32_t off, uint32_t a) {
  asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(off), "v"(a), "s"(base) : "memory");
}

// The row walker is bound by its vector ALU, so its arithmetic is the formulas above reduced to
// what the hardware does at full rate -- identical results, proven here, checked by the parity
// tests for both models:
//  * products through the 24-bit multiplier (coefficients < 2^15, channel sums < 2^10, 15-bit
//    chroma times taps <= 2048: every operand fits);
//  * luma: clip8((min(2 Y14, 32767) + 64) >> 7) with Y14 = S >> 9 is min((S + 16384) >> 15, 255):
//    (2a + 64) >> 7 = (a + 32) >> 6 and floor((floor(S / 512) + 32) / 64) = floor((S + 16384) /
//    32768); once 2 Y14 exceeds 32767 both forms give 255; S > 0, so no lower clamp;
//  * chroma: U14 <= (BU * 510 + (256 << 15) + 512) >> 10 = 15360, so min(2 U14, 32767) never
//    clamps (V likewise: RV = BU), and U14 >= 0.
__device__ __forceinline__ uint32_t r2y_luma8(const Rgb2YuvConsts &k, uint32_t px) {
  const int r = px & 0xff, g = (px >> 8) & 0xff, b = (px >> 16) & 0xff;
  const int s = __mul24(k.ry, r) + __mul24(k.gy, g) + __mul24(k.by, b) + ((32 << 14) + (1 << 8) + 16384);
  return min((uint32_t)s >> 15, 255u);
}
__device__ __forceinline__ int2 r2y_chroma15(const Rgb2YuvConsts &k, uint32_t p0, uint32_t p1) {
  const uint32_t rb = (p0 & 0x00ff00ffu) + (p1 & 0x00ff00ffu);  // R sum | B sum << 16
  const int r = rb & 0xffff, b = rb >> 16;
  const int g = (int)((p0 >> 8) & 0xff) + (int)((p1 >> 8) & 0xff);
  const int su = __mul24(k.ru, r) + __mul24(k.gu, g) + __mul24(k.bu, b) + ((256 << 15) + (1 << 9));
  const int sv = __mul24(k.rv, r) + __mul24(k.gv, g) + __mul24(k.bv, b) + ((256 << 15) + (1 << 9));
  return make_int2((su >> 10) << 1, (sv >> 10) << 1);
}

struct R2yRow {
  uint32_t px[8];
};
__device__ __forceinline__ R2yRow r2y_load(const uint8_t *src, int src_linesize, int y,
                                           uint32_t xoff) {
  typedef uint32_t u32x4_r __attribute__((ext_vector_type(4)));
  const uint8_t *row = src + (size_t)y * src_linesize + xoff;
  const u32x4_r a = *reinterpret_cast<const u32x4_r *>(row);
  const u32x4_r b = *reinterpret_cast<const u32x4_r *>(row + 16);
  return R2yRow{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}

template <int MODEL>
__global__ __launch_bounds__(256) void rgb0_to_yuv420p_walk_kernel(
    uint8_t *__restrict__ y_dst, uint8_t *__restrict__ u_dst, uint8_t *__restrict__ v_dst,
    int y_linesize, int u_linesize, int v_linesize, const uint8_t *__restrict__ src,
    int src_linesize, int width, int height, const Rgb2YuvConsts k, int rows_per_wave,
    int nstrips) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int strip = wave % nstrips;
  const int cy0 = (wave / nstrips) * rows_per_wave;
  const int cw = width >> 1, ch = height >> 1;
  if (cy0 >= ch) return;
  const int cy1 = min(cy0 + rows_per_wave, ch);
  const int c_own = (strip * 64 + lane) * 4;  // first chroma column of the lane
  const bool writes = c_own < cw;             // width % 8 == 0: a lane is in or out as a whole
  const int c0 = min(c_own, cw - 4);          // idle lanes read in bounds and store nothing
  const uint32_t xoff = (uint32_t)c0 * 8u;
  const int y_last = height - 1;

  int wu[4][4], wv[4][4];  // window [slot][pair]: chroma of source rows 2cy - 1 .. 2cy + 2
  // one source row: its luma (stored if the row is this wave's) and its chroma into a slot
  auto take_row = [&](const R2yRow &r, int y, int slot, bool own_luma) {
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t yv = r2y_luma8(k, r.px[q]);
      if (q < 4) w0 |= yv << (8 * q);
      else w1 |= yv << (8 * (q - 4));
    }
    if (own_luma && writes)
      r2y_store8(y_dst, (uint32_t)y * (uint32_t)y_linesize + (uint32_t)c0 * 2u, w0, w1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int2 c = r2y_chroma15(k, r.px[2 * q], r.px[2 * q + 1]);
      wu[slot][q] = c.x;
      wv[slot][q] = c.y;
    }
  };
  // prologue: rows 2cy0 - 1 (clamped; never this wave's luma) and 2cy0
  {
    const R2yRow a = r2y_load(src, src_linesize, max(2 * cy0 - 1, 0), xoff);
    const R2yRow b = r2y_load(src, src_linesize, 2 * cy0, xoff);
    take_row(a, 0, 0, false);
    take_row(b, 2 * cy0, 1, true);
  }
  R2yRow na = r2y_load(src, src_linesize, min(2 * cy0 + 1, y_last), xoff);
  R2yRow nb = r2y_load(src, src_linesize, min(2 * cy0 + 2, y_last), xoff);
  for (int cy = cy0; cy < cy1; ++cy) {
    const R2yRow ra = na, rb = nb;
    // the next chroma row's two source rows, unconditionally (clamped addresses)
    na = r2y_load(src, src_linesize, min(2 * cy + 3, y_last), xoff);
    nb = r2y_load(src, src_linesize, min(2 * cy + 4, y_last), xoff);
    take_row(ra, 2 * cy + 1, 2, true);
    // row 2cy + 2 is luma row 0 of chroma row cy + 1: this wave's unless that is the next run's
    take_row(rb, 2 * cy + 2, 3, cy + 1 < cy1);
    // taps on the window's rows: interior 512 1536 1536 512; folded at the frame's borders
    // (first chroma row: 0 2048 1536 512 -- row -1 was loaded clamped and carries no weight --,
    // last: 512 1536 2048 0).  x86 builds: every chroma row but the last through the truncating
    // 16-bit multiply (pmulhw: (c * tap) >> 16 per tap, paddw wraps; the sum of the terms wraps
    // to the same value).  Interior rows take the taps as constants: (c * 512) >> 16 = c >> 7,
    // (c * 1536) >> 16 = (3 c) >> 7.
    uint32_t ub = 0, vb = 0;
    if (cy > 0 && cy < ch - 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MODEL == 1) {
          const int au = 5 + (wu[0][q] >> 7) + ((3 * wu[1][q]) >> 7) + ((3 * wu[2][q]) >> 7) +
                         (wu[3][q] >> 7);
          const int av = 5 + (wv[0][q] >> 7) + ((3 * wv[1][q]) >> 7) + ((3 * wv[2][q]) >> 7) +
                         (wv[3][q] >> 7);
          ub |= shift_clip8<3>((int)(int16_t)au) << (8 * q);
          vb |= shift_clip8<3>((int)(int16_t)av) << (8 * q);
        } else {
          const int au = (64 << 12) + 512 * (wu[0][q] + wu[3][q]) + 1536 * (wu[1][q] + wu[2][q]);
          const int av = (64 << 12) + 512 * (wv[0][q] + wv[3][q]) + 1536 * (wv[1][q] + wv[2][q]);
          ub |= shift_clip8<19>(au) << (8 * q);
          vb |= shift_clip8<19>(av) << (8 * q);
        }
      }
    } else {
      int t0 = 512, t1 = 1536, t2 = 1536, t3 = 512;
      if (cy == 0) {
        t0 = 0;
        t1 = 2048;
      }
      if (cy == ch - 1) {
        t2 = 2048;
        t3 = 0;
      }
      const bool mmx = MODEL == 1 && cy < ch - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (mmx) {
          const int au = 5 + (__mul24(wu[0][q], t0) >> 16) + (__mul24(wu[1][q], t1) >> 16) +
                         (__mul24(wu[2][q], t2) >> 16) + (__mul24(wu[3][q], t3) >> 16);
          const int av = 5 + (__mul24(wv[0][q], t0) >> 16) + (__mul24(wv[1][q], t1) >> 16) +
                         (__mul24(wv[2][q], t2) >> 16) + (__mul24(wv[3][q], t3) >> 16);
          ub |= shift_clip8<3>((int)(int16_t)au) << (8 * q);
          vb |= shift_clip8<3>((int)(int16_t)av) << (8 * q);
        } else {
          const int au = (64 << 12) + __mul24(wu[0][q], t0) + __mul24(wu[1][q], t1) +
                         __mul24(wu[2][q], t2) + __mul24(wu[3][q], t3);
          const int av = (64 << 12) + __mul24(wv[0][q], t0) + __mul24(wv[1][q], t1) +
                         __mul24(wv[2][q], t2) + __mul24(wv[3][q], t3);
          ub |= shift_clip8<19>(au) << (8 * q);
          vb |= shift_clip8<19>(av) << (8 * q);
        }
      }
    }
    if (writes) {
      r2y_store4(u_dst, (uint32_t)cy * (uint32_t)u_linesize + (uint32_t)c0, ub);
      r2y_store4(v_dst, (uint32_t)cy * (uint32_t)v_linesize + (uint32_t)c0, vb);
    }
    // slide by two source rows
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wu[0][q] = wu[2][q];
      wv[0][q] = wv[2][q];
      wu[1][q] = wu[3][q];
      wv[1][q] = wv[3][q];
    }
  }
}

}  // namespace

extern "C" int f360_rgb0_to_yuv420p(f360_ctx *ctx, uint8_t *y_dev, uint8_t *u_dev, uint8_t *v_dev,
                                    int y_linesize, int u_linesize, int v_linesize,
                                    const uint8_t *src_dev, int src_linesize, int width,
                                    int height) {
  F360_REQUIRE(ctx, "f360_rgb0_to_yuv420p: null context");
  F360_REQUIRE(y_dev && u_dev && v_dev && src_dev, "f360_rgb0_to_yuv420p: null buffer");
  // odd sizes change libswscale's chroma geometry (ceil'ed plane sizes, a non-integer vertical
  // step); below 8 rows initFilter shortens the vertical filter: neither is provided
  F360_REQUIRE(width >= 2 && height >= 8 && width % 2 == 0 && height % 2 == 0,
               "f360_rgb0_to_yuv420p: bad size %dx%d (even width, even height >= 8)", width,
               height);
  F360_REQUIRE(src_linesize >= 4 * width && y_linesize >= width && u_linesize >= width / 2 &&
                   v_linesize >= width / 2,
               "f360_rgb0_to_yuv420p: linesize too small");
  F360_BIND_DEVICE(ctx);
  const Rgb2YuvConsts k = rgb2yuv_consts();
  const bool vec = width % 8 == 0 && src_linesize % 16 == 0 && y_linesize % 8 == 0 &&
                   u_linesize % 4 == 0 && v_linesize % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(src_dev) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y_dev) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(u_dev) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(v_dev) & 3) == 0;
  const bool prof = f360::take_profile_slot(ctx);
  f360::KernelSpan span(ctx, f360::kRgbToYuv, prof);
  const bool x86 = ctx->opt_yuv_model == 1;
  const int cw = width / 2, ch = height / 2;
#define F360_R2Y(M, P)                                                                          \
  hipLaunchKernelGGL((rgb0_to_yuv420p_kernel<M, P>), dim3(((cw + P - 1) / P + 63) / 64, (ch + 3) / 4), \
                     dim3(256), 0, ctx->stream, y_dev, u_dev, v_dev, y_linesize, u_linesize,    \
                     v_linesize, src_dev, src_linesize, width, height, k)
  // the row-walking kernel: vector layout, 32-bit store offsets and enough waves.
  // "yuv.r2y_rows" = 0 (default): the longest runs of 16 / 8 / 4 chroma rows that still give
  // 1536 waves, 4 down to 512 waves (8K: 16 rows, 42.2 -> 34.4 us; the 4272x2144 reduced frame:
  // 4 rows, 16.4 -> 14.0 us), below that the kernel with one chroma row per thread; n > 0:
  // always, with runs of n; -1: never
  const int nstrips = (cw / 4 + 63) / 64;
  auto waves_with = [&](int r) { return (long)nstrips * ((ch + r - 1) / r); };
  int rpw = ctx->opt_r2y_rows;
  if (rpw == 0) rpw = waves_with(16) >= 1536 ? 16 : waves_with(8) >= 1536 ? 8 : waves_with(4) >= 512 ? 4 : -1;
  const long waves = rpw > 0 ? waves_with(rpw) : 0;
  if (vec && rpw > 0 && (size_t)y_linesize * height < ((size_t)1 << 32)) {
    const dim3 wgrid((unsigned)((waves + 3) / 4));
    if (x86)
      hipLaunchKernelGGL(rgb0_to_yuv420p_walk_kernel<1>, wgrid, dim3(256), 0, ctx->stream, y_dev,
                         u_dev, v_dev, y_linesize, u_linesize, v_linesize, src_dev, src_linesize,
                         width, height, k, rpw, nstrips);
    else
      hipLaunchKernelGGL(rgb0_to_yuv420p_walk_kernel<0>, wgrid, dim3(256), 0, ctx->stream, y_dev,
                         u_dev, v_dev, y_linesize, u_linesize, v_linesize, src_dev, src_linesize,
                         width, height, k, rpw, nstrips);
  } else if (vec) {
    if (x86) F360_R2Y(1, 4);
    else F360_R2Y(0, 4);
  } else {
    if (x86) F360_R2Y(1, 1);
    else F360_R2Y(0, 1);
  }
#undef F360_R2Y
  F360_HIP_TRY(hipGetLastError());
  return F360_OK;
}
