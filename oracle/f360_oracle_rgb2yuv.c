/*
 * f360_oracle_rgb2yuv.c -- CPU oracle for the colour-space step BEHIND the hot path
 * (output side).  TEST INFRASTRUCTURE ONLY, see f360_oracle.h.
 *
 * PARITY UNPINNED: libswscale cannot be built or run in this container (FFmpeg's
 * configure-generated config.h / avconfig.h are not vendored), so nothing here has been
 * compared with a running sws_scale; it restates the vendored FFmpeg 4.2 source text.
 *
 * What the reference does.  VideoEncoder::EncodeFrame turns the reduced RGB0 frame into
 * yuv420p in front of NVENC with
 *   sws_getContext(w, h, AV_PIX_FMT_RGB0, w, h, AV_PIX_FMT_YUV420P, SWS_BILINEAR, ...) + sws_scale
 *   (src/video_encoder.cc:380-395).
 * Inside FFmpeg 4.2 (paths relative to include/FFmpeg42/libswscale/):
 *   - RGB0 is treated as RGBA, alpha ignored                       utils.c:1049-1058,1071
 *   - no unscaled special converter exists for RGBA -> yuv420p (only BGR24 has one,
 *     swscale_unscaled.c), so the GENERIC scaler runs
 *   - RGB input: chroma is taken from pixel PAIRS (chrSrcHSubSample = 1, utils.c:1385-1405)
 *     but from EVERY source row (chrSrcVSubSample = 0); the destination halves it vertically
 *   - input conversion, little endian RGBA == AV_PIX_FMT_BGR32: lumToYV12 = rgb32ToY_c,
 *     chrToYV12 = rgb32ToUV_half_c          input.c:252-275,304-346,380,1213-1216,1537-1541
 *     (x86 builds use ff_rgbaToY_sse2 for the luma: same integers, x86/input.asm rgb_Yrnd)
 *     with the ITU-R 601 limited-range coefficients              utils.c:811-821
 *   - horizontal "scaling" with a one-tap filter of 1 << 14: hScale16To15_c shifts by 13
 *     for RGB sources, i.e. doubles the 14-bit sample          swscale.c:95-121, utils.c:352-361
 *   - vertical: luma one tap (yuv2plane1_8_c, output.c), chroma the bilinear 2:1 filter
 *     initFilter builds (utils.c:331-726, restated below as it stands), applied by
 *     yuv2planeX_8_c (output.c) -- or, on x86 builds without SWS_ACCURATE_RND, by the inline
 *     MMX/SSE yuv2yuvX whose pmulhw truncates every tap's product (x86/swscale.c:201-275), for
 *     all rows but the last one (swscale.c: dstY >= dstH - 2 falls back to the C functions).
 * Two models again, as on the input side:
 *   F360O_YUV_SWS_C    the C functions throughout
 *   F360O_YUV_SWS_X86  the x86 vertical chroma scaler for chroma rows 0 .. h/2 - 2
 * Width and height must be even (the reference's reduced sizes are multiples of 16).
 */
#include "f360_oracle.h"

#include <stdlib.h>
#include <string.h>

#define RGB2YUV_SHIFT 15 /* swscale_internal.h:415 */

/* utils.c:811-821, the SWS_CS_DEFAULT (ITU601) special case of fill_rgb2yuv_table */
static const int RY = (int)(0.299 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int GY = (int)(0.587 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int BY = (int)(0.114 * 219 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int RU = -(int)(0.169 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int GU = -(int)(0.331 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int BU = (int)(0.500 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int RV = (int)(0.500 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int GV = -(int)(0.419 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);
static const int BV = -(int)(0.081 * 224 / 255 * (1 << RGB2YUV_SHIFT) + 0.5);

void f360o_rgb2yuv_coeffs(int32_t *out9) {
  const int c[9] = {RY, GY, BY, RU, GU, BU, RV, GV, BV};
  for (int k = 0; k < 9; ++k) out9[k] = c[k];
}

static uint32_t load_le32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* rgb32ToY_c: rgb16_32ToY_c_template(shr 0, shg 0, shb 16, shp 0, maskr 0xFF, maskg 0xFF00,
 * maskb 0xFF0000, rsh 8, gsh 0, bsh 8, S = RGB2YUV_SHIFT + 8), input.c:252-275,380 */
static void rgb32_to_y(int16_t *dst, const uint8_t *src, int width) {
  const int S = RGB2YUV_SHIFT + 8;
  const int ry = RY << 8, gy = GY << 0, by = BY << 8;
  const unsigned rnd = (32u << (S - 1)) + (1u << (S - 7));
  for (int i = 0; i < width; ++i) {
    const int px = (int)load_le32(src + 4 * i);
    const int b = (px & 0xFF0000) >> 16;
    const int g = (px & 0xFF00) >> 0;
    const int r = (px & 0x00FF) >> 0;
    dst[i] = (int16_t)((ry * r + gy * g + by * b + rnd) >> (S - 6));
  }
}

/* rgb32ToUV_half_c: rgb16_32ToUV_half_c_template, input.c:304-346 */
static void rgb32_to_uv_half(int16_t *dst_u, int16_t *dst_v, const uint8_t *src, int width) {
  const int S = RGB2YUV_SHIFT + 8;
  const int ru = RU * (1 << 8), gu = GU * (1 << 0), bu = BU * (1 << 8);
  const int rv = RV * (1 << 8), gv = GV * (1 << 0), bv = BV * (1 << 8);
  int maskr = 0x00FF, maskg = 0xFF00, maskb = 0xFF0000;
  const int maskgx = ~(maskr | maskb);
  const unsigned rnd = (256u << S) + (1u << (S - 6));
  maskr |= maskr << 1;
  maskb |= maskb << 1;
  maskg |= maskg << 1;
  for (int i = 0; i < width; ++i) {
    const unsigned px0 = load_le32(src + 4 * (2 * i + 0));
    const unsigned px1 = load_le32(src + 4 * (2 * i + 1));
    int g = (int)((px0 & (unsigned)maskgx) + (px1 & (unsigned)maskgx));
    const int rb = (int)(px0 + px1 - (unsigned)g);
    const int b = (rb & maskb) >> 16;
    g = (g & maskg) >> 0;
    const int r = (rb & maskr) >> 0;
    dst_u[i] = (int16_t)(((unsigned)(ru * r + gu * g + bu * b) + rnd) >> (S - 6 + 1));
    dst_v[i] = (int16_t)(((unsigned)(rv * r + gv * g + bv * b) + rnd) >> (S - 6 + 1));
  }
}

/* hScale16To15_c with the unscaled one-tap filter (1 << 14) and sh = 13 (RGB source),
 * swscale.c:95-121 */
static void hscale_unscaled(int16_t *dst, const int16_t *src, int n) {
  for (int i = 0; i < n; ++i) {
    const int val = (int)(uint16_t)src[i] * (1 << 14);
    const int v = val >> 13;
    dst[i] = (int16_t)(v < (1 << 15) - 1 ? v : (1 << 15) - 1);
  }
}

/* initFilter (utils.c:331-726) for flags = SWS_BILINEAR, no source / destination filter
 * vectors, restated as it stands.  Returns the filter size; *filter_out (dst_w * size int16)
 * and *pos_out (dst_w int32) are malloc'ed. */
#define ROUNDED_DIV(a, b) (((a) >= 0 ? (a) + ((b) >> 1) : (a) - ((b) >> 1)) / (b))
static int ilog2(unsigned v) {
  int n = 0;
  while (v >>= 1) ++n;
  return n;
}
static int sws_init_filter_bilinear(int16_t **filter_out, int32_t **pos_out, int x_inc, int src_w,
                                    int dst_w, int filter_align, int one, int src_pos,
                                    int dst_pos) {
  const int l2 = ilog2((unsigned)(src_w / dst_w));
  const int64_t fone = 1LL << (54 - (l2 < 8 ? l2 : 8));
  int filter_size;
  int64_t *filter;
  int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(dst_w + 3));
  if (llabs((long long)x_inc - 0x10000) < 10 && src_pos == dst_pos) { /* unscaled, :352-361 */
    filter_size = 1;
    filter = (int64_t *)calloc((size_t)dst_w, sizeof(int64_t));
    for (int i = 0; i < dst_w; ++i) {
      filter[i] = fone;
      pos[i] = i;
    }
  } else { /* :401-510, SWS_BILINEAR: size factor 2 */
    const int size_factor = 2;
    if (x_inc <= 1 << 16)
      filter_size = 1 + size_factor;
    else
      filter_size = 1 + (size_factor * src_w + dst_w - 1) / dst_w;
    if (filter_size > src_w - 2) filter_size = src_w - 2;
    if (filter_size < 1) filter_size = 1;
    filter = (int64_t *)malloc(sizeof(int64_t) * (size_t)dst_w * filter_size);
    int64_t x_dst_in_src = ((dst_pos * (int64_t)x_inc) >> 7) - ((src_pos * 0x10000LL) >> 7);
    for (int i = 0; i < dst_w; ++i) {
      int xx = (int)((x_dst_in_src - (filter_size - 2) * (1LL << 16)) / (1 << 17));
      pos[i] = xx;
      for (int j = 0; j < filter_size; ++j) {
        int64_t d = (llabs(((int64_t)xx * (1 << 17)) - x_dst_in_src)) << 13;
        if (x_inc > 1 << 16) d = d * dst_w / src_w;
        int64_t coeff = (1 << 30) - d;
        if (coeff < 0) coeff = 0;
        coeff *= fone >> 30;
        filter[(size_t)i * filter_size + j] = coeff;
        xx++;
      }
      x_dst_in_src += 2 * x_inc;
    }
  }
  /* no source / destination filter: filter2 == filter; filterPos unchanged (:512-541) */
  const int filter2_size = filter_size;
  int64_t *filter2 = filter;
  /* reduce the filter size, step 1 (:543-580) */
  int min_filter_size = 0;
  for (int i = dst_w - 1; i >= 0; --i) {
    int min = filter2_size;
    int64_t cut_off = 0;
    for (int j = 0; j < filter2_size; ++j) {
      cut_off += llabs(filter2[(size_t)i * filter2_size]);
      if ((double)cut_off > 0.002 * (double)fone) break; /* SWS_MAX_REDUCE_CUTOFF */
      if (i < dst_w - 1 && pos[i] >= pos[i + 1]) break;
      for (int k = 1; k < filter2_size; ++k)
        filter2[(size_t)i * filter2_size + k - 1] = filter2[(size_t)i * filter2_size + k];
      filter2[(size_t)i * filter2_size + filter2_size - 1] = 0;
      pos[i]++;
    }
    cut_off = 0;
    for (int j = filter2_size - 1; j > 0; --j) {
      cut_off += llabs(filter2[(size_t)i * filter2_size + j]);
      if ((double)cut_off > 0.002 * (double)fone) break;
      min--;
    }
    if (min > min_filter_size) min_filter_size = min;
  }
  if (min_filter_size == 1 && filter_align == 2) filter_align = 1; /* :597-601, MMX builds */
  const int out_size = (min_filter_size + (filter_align - 1)) & ~(filter_align - 1);
  int64_t *f = (int64_t *)malloc(sizeof(int64_t) * (size_t)dst_w * out_size);
  for (int i = 0; i < dst_w; ++i)
    for (int j = 0; j < out_size; ++j)
      f[(size_t)i * out_size + j] = j >= filter2_size ? 0 : filter2[(size_t)i * filter2_size + j];
  /* fix borders (:634-676) */
  for (int i = 0; i < dst_w; ++i) {
    if (pos[i] < 0) {
      for (int j = 1; j < out_size; ++j) {
        const int left = j + pos[i] > 0 ? j + pos[i] : 0;
        f[(size_t)i * out_size + left] += f[(size_t)i * out_size + j];
        f[(size_t)i * out_size + j] = 0;
      }
      pos[i] = 0;
    }
    if (pos[i] + out_size > src_w) {
      const int shift = pos[i] + (out_size - src_w < 0 ? out_size - src_w : 0);
      int64_t acc = 0;
      for (int j = out_size - 1; j >= 0; --j)
        if (pos[i] + j >= src_w) {
          acc += f[(size_t)i * out_size + j];
          f[(size_t)i * out_size + j] = 0;
        }
      for (int j = out_size - 1; j >= 0; --j)
        f[(size_t)i * out_size + j] = j < shift ? 0 : f[(size_t)i * out_size + j - shift];
      pos[i] -= shift;
      f[(size_t)i * out_size + src_w - 1 - pos[i]] += acc;
    }
  }
  /* normalise with error diffusion (:683-703) */
  int16_t *out = (int16_t *)calloc((size_t)(dst_w + 3) * out_size, sizeof(int16_t));
  for (int i = 0; i < dst_w; ++i) {
    int64_t error = 0, sum = 0;
    for (int j = 0; j < out_size; ++j) sum += f[(size_t)i * out_size + j];
    sum = (sum + one / 2) / one;
    if (!sum) sum = 1;
    for (int j = 0; j < out_size; ++j) {
      const int64_t v = f[(size_t)i * out_size + j] + error;
      const int int_v = (int)ROUNDED_DIV(v, sum);
      out[(size_t)i * out_size + j] = (int16_t)int_v;
      error = v - int_v * sum;
    }
  }
  free(filter);
  free(f);
  *filter_out = out;
  *pos_out = pos;
  return out_size;
}

/* get_local_pos, utils.c:302-309 */
static int get_local_pos(int chr_subsample, int pos) {
  if (pos == -1 || pos <= -513) pos = (128 << chr_subsample) - 128;
  pos += 128;
  return pos >> chr_subsample;
}

int f360o_rgb2yuv_chroma_vfilter(int16_t *filter_out, int32_t *pos_out, int height, int max_size) {
  const int chr_src_h = height, chr_dst_h = (height + 1) >> 1;
  const int chr_y_inc = (int)((((int64_t)chr_src_h << 16) + (chr_dst_h >> 1)) / chr_dst_h);
  int16_t *f;
  int32_t *p;
  /* filterAlign 2 on x86 MMX builds, 1 otherwise (utils.c:1699-1701): the size comes out as 4
   * either way for a 2:1 bilinear filter */
  const int size = sws_init_filter_bilinear(&f, &p, chr_y_inc, chr_src_h, chr_dst_h, 2, 1 << 12,
                                            get_local_pos(0, -513), get_local_pos(1, -513));
  if (size <= max_size) {
    memcpy(filter_out, f, sizeof(int16_t) * (size_t)chr_dst_h * size);
    memcpy(pos_out, p, sizeof(int32_t) * (size_t)chr_dst_h);
  }
  free(f);
  free(p);
  return size;
}

static uint8_t clip_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void f360o_rgb0_to_yuv420p(uint8_t *y_plane, int y_linesize, uint8_t *u_plane, int u_linesize,
                           uint8_t *v_plane, int v_linesize, const uint8_t *src, int src_linesize,
                           int width, int height, int model) {
  const int cw = width >> 1, ch = height >> 1;
  int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)width);
  int16_t *y15 = (int16_t *)malloc(sizeof(int16_t) * (size_t)width);
  int16_t *u15 = (int16_t *)malloc(sizeof(int16_t) * (size_t)cw * height);
  int16_t *v15 = (int16_t *)malloc(sizeof(int16_t) * (size_t)cw * height);
  int16_t *tu = (int16_t *)malloc(sizeof(int16_t) * (size_t)cw);
  int16_t *tv = (int16_t *)malloc(sizeof(int16_t) * (size_t)cw);
  for (int y = 0; y < height; ++y) {
    const uint8_t *row = src + (size_t)y * src_linesize;
    /* luma: input conversion, one-tap horizontal filter, yuv2plane1_8_c with the constant
     * dither 64 of 8-bit sources (output.c; ff_sws_pb_64) */
    rgb32_to_y(tmp, row, width);
    hscale_unscaled(y15, tmp, width);
    for (int x = 0; x < width; ++x) y_plane[(size_t)y * y_linesize + x] = clip_u8((y15[x] + 64) >> 7);
    rgb32_to_uv_half(tu, tv, row, cw);
    hscale_unscaled(u15 + (size_t)y * cw, tu, cw);
    hscale_unscaled(v15 + (size_t)y * cw, tv, cw);
  }
  int16_t *vf = (int16_t *)malloc(sizeof(int16_t) * (size_t)ch * 8);
  int32_t *vp = (int32_t *)malloc(sizeof(int32_t) * (size_t)ch);
  const int fs = f360o_rgb2yuv_chroma_vfilter(vf, vp, height, 8);
  for (int cy = 0; cy < ch; ++cy) {
    /* x86: yuv2yuvX for every chroma row computed at dstY < dstH - 2 */
    const int mmx = model == F360O_YUV_SWS_X86 && cy < ch - 1;
    for (int plane = 0; plane < 2; ++plane) {
      const int16_t *s15 = plane ? v15 : u15;
      uint8_t *dst = (plane ? v_plane : u_plane) + (size_t)cy * (plane ? v_linesize : u_linesize);
      for (int x = 0; x < cw; ++x) {
        if (mmx) {
          /* x86/swscale.c:209-248: start value ((dither + ((filterSize - 1) << 3)) >> 4) in
           * 16-bit lanes, one pmulhw per tap, arithmetic shift by 3, packuswb */
          int acc = (64 + ((fs - 1) << 3)) >> 4;
          for (int j = 0; j < fs; ++j)
            acc = (int16_t)(acc + (int16_t)(((int)s15[(size_t)(vp[cy] + j) * cw + x] * vf[cy * fs + j]) >> 16));
          dst[x] = clip_u8(acc >> 3);
        } else {
          /* yuv2planeX_8_c, output.c */
          int val = 64 << 12;
          for (int j = 0; j < fs; ++j) val += s15[(size_t)(vp[cy] + j) * cw + x] * vf[cy * fs + j];
          dst[x] = clip_u8(val >> 19);
        }
      }
    }
  }
  free(tmp);
  free(y15);
  free(u15);
  free(v15);
  free(tu);
  free(tv);
  free(vf);
  free(vp);
}
