// yuv_device.h -- YUV (ITU-R 601, limited range) -> RGB arithmetic of libswscale's unscaled
// yuv420p -> RGB converter, which is what the reference's VideoDecoder runs in front of the
// hot path (src/video_decoder.cc:167-170,222-224; FFmpeg 4.2, include/FFmpeg42/libswscale/).
// Two arithmetic models, because libswscale has two converters that round differently:
//   model 0  the table-driven C converter yuv2rgb_c_32 (yuv2rgb.c:70-81,241-262, tables
//            :774-855,968-993).  The table is a clipped affine function of its index, so the
//            lookup is evaluated in closed form:
//              value = clip8((c0 + (Y + off(U,V)) * cy) >> 16)
//   model 1  the x86 MMX converter (x86/yuv2rgb_template.c:84-122): 16-bit fixed point with
//            pmulhw.  None of its saturating adds can saturate for 8-bit inputs.
// Chroma is the sample of the 2x2 block, not interpolated, in both.
// The constants come from f360::build_yuv2rgb_consts (host_tables.cpp).
#pragma once

#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace f360 {

struct YuvConsts {
  // model 0
  int cy, c0;              // luma step and offset of the clipped ramp (16.16)
  int crv, cbu, cgu, cgv;  // chroma increments after the division by cy (yuv2rgb.c:846-850)
  int r0, gu0, gv0, b0;    // -(inc >> 9): table offset of the neutral chroma value 128
  // model 1
  int yc, vrc, ubc, vgc, ugc, yoff;  // 16-bit lanes of c->yCoeff ... c->yOffset
};

struct YuvPlanes {
  const uint8_t *y, *u, *v;
  int y_linesize, u_linesize, v_linesize;
};

#if defined(__HIPCC__)
// What one chroma sample adds to R, G and B of the (up to four) pixels that share it.
struct ChromaTerms {
  int r, g, b;
};

// Plain 32-bit products: forcing the 24-bit multiplier (__mul24; every factor fits) measured
// 7 % slower in the table writer, the compiler already picks v_mad_u32_u24 where it can prove
// the ranges.
__device__ __forceinline__ int mul24(int a, int b) { return a * b; }

template <int MODEL>
__device__ __forceinline__ ChromaTerms chroma_terms(const YuvConsts &k, int U, int V) {
  ChromaTerms t;
  if (MODEL == 0) {
    t.r = mul24(k.r0 + (mul24(V, k.crv) >> 16), k.cy);
    t.g = mul24(k.gu0 + (mul24(U, k.cgu) >> 16) + k.gv0 + (mul24(V, k.cgv) >> 16), k.cy);
    t.b = mul24(k.b0 + (mul24(U, k.cbu) >> 16), k.cy);
  } else {
    const int u = (U << 3) - 0x400, v = (V << 3) - 0x400;
    t.r = mul24(v, k.vrc) >> 16;
    t.g = (mul24(u, k.ugc) >> 16) + (mul24(v, k.vgc) >> 16);
    t.b = mul24(u, k.ubc) >> 16;
  }
  return t;
}

__device__ __forceinline__ uint32_t clip8(int v) { return (uint32_t)min(max(v, 0), 255); }
// clip8(v >> 16), written as clamp-then-shift on purpose: hipcc 7.2 turns two adjacent
// "arithmetic shift, clamp to 0..255" results into one v_ashr_pk_u8_i32 and then ORs the third
// byte into that register as if its bits 31:16 were zero, which they are not on gfx950 (the
// blue channel came out OR-ed with garbage).  This form does not match that pattern.
__device__ __forceinline__ uint32_t clip8_shr16(int v) {
  return (uint32_t)min(max(v, 0), 0x00ffffff) >> 16;
}

// One pixel as R | G << 8 | B << 16.
template <int MODEL>
__device__ __forceinline__ uint32_t yuv_pixel(const YuvConsts &k, int Y, const ChromaTerms &t) {
  if (MODEL == 0) {
    const int base = k.c0 + mul24(Y, k.cy);
    return clip8_shr16(base + t.r) | (clip8_shr16(base + t.g) << 8) |
           (clip8_shr16(base + t.b) << 16);
  }
  const int yy = mul24((Y << 3) - k.yoff, k.yc) >> 16;
  return clip8(yy + t.r) | (clip8(yy + t.g) << 8) | (clip8(yy + t.b) << 16);
}

// Four pixels of one row: `y4` holds their luma bytes, `uv` the two chroma pairs as
// U0 | U1 << 8 | V0 << 16 | V1 << 24.
template <int MODEL>
__device__ __forceinline__ void yuv_pixels4(const YuvConsts &k, uint32_t y4, uint32_t uv,
                                            uint32_t (&px)[4]) {
  const ChromaTerms t0 = chroma_terms<MODEL>(k, (int)(uv & 0xffu), (int)((uv >> 16) & 0xffu));
  const ChromaTerms t1 = chroma_terms<MODEL>(k, (int)((uv >> 8) & 0xffu), (int)(uv >> 24));
  px[0] = yuv_pixel<MODEL>(k, (int)(y4 & 0xffu), t0);
  px[1] = yuv_pixel<MODEL>(k, (int)((y4 >> 8) & 0xffu), t0);
  px[2] = yuv_pixel<MODEL>(k, (int)((y4 >> 16) & 0xffu), t1);
  px[3] = yuv_pixel<MODEL>(k, (int)(y4 >> 24), t1);
}
#endif  // __HIPCC__

}  // namespace f360
