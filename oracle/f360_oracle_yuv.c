/*
 * f360_oracle_yuv.c -- CPU oracle for the colour-space step in front of the hot path
 * (SURVEY.md 8f-3).  TEST INFRASTRUCTURE ONLY, see f360_oracle.h.
 *
 * PARITY UNPINNED: libswscale cannot be built or run in this container (FFmpeg's
 * configure-generated config.h / avconfig.h are not vendored), so nothing here has been
 * compared with a running sws_scale; it restates the vendored FFmpeg 4.2 source text.
 *
 * What the reference does.  VideoDecoder::GetFrame turns every decoded frame into RGB0 with
 *   sws_getContext(w, h, yuv420p, w, h, AV_PIX_FMT_RGB0, SWS_BILINEAR, ...) + sws_scale
 *   (src/video_decoder.cc:167-170,222-224).
 * Inside FFmpeg 4.2 (paths below are relative to include/FFmpeg42/libswscale/):
 *   - RGB0 is treated as RGBA with a constant alpha          utils.c:1049-1058,1071-1072
 *   - colour matrix SWS_CS_DEFAULT = ITU-R 601, limited-range source, brightness 0,
 *     contrast = saturation = 1<<16                          utils.c:1200-1203, swscale.h:91-95
 *   - same size, no filters -> ff_get_unscaled_swscale       utils.c:1817-1829
 *   - yuv420p -> any RGB, even height, no SWS_ACCURATE_RND -> ff_yuv2rgb_get_func_ptr
 *                                                            swscale_unscaled.c:1931-1936
 *   - that returns the x86 MMX converter when the build has inline asm
 *     (yuv2rgb.c:679-688, x86/yuv2rgb.c: AV_PIX_FMT_BGR32 == RGBA on little endian ->
 *     yuv420_bgr32_mmx), otherwise the table-driven C converter yuv2rgb_c_32
 *     (yuv2rgb.c:705-708).
 * The two converters round differently, so "what sws_scale returns" depends on the build.
 * Both are restated here:
 *   F360O_YUV_SWS_C    yuv2rgb_c_32, yuv2rgb.c:70-81,129-172,241-262 with the tables of
 *                      ff_yuv2rgb_c_init_tables, yuv2rgb.c:737-760,774-855,968-993
 *   F360O_YUV_SWS_X86  the YUV2RGB / RGB_PACK32 macros of x86/yuv2rgb_template.c:84-122,
 *                      356-378,424-443 with the 16-bit coefficients of yuv2rgb.c:762-772,
 *                      830-837
 * Both use the chroma sample of the 2x2 block unchanged (no chroma interpolation) and write
 * 255 into the fourth byte.  Every pixel x < width is converted; the reference leaves the last
 * width%8 (x86) / an odd last (C) pixel of a row untouched, which no frame size the reference
 * handles (1920, 3840, 7680 wide) exercises.
 */
#include "f360_oracle.h"

#include <stdlib.h>
#include <string.h>

#define HEADROOM 512      /* YUVRGB_TABLE_HEADROOM, swscale_internal.h:38 */
#define LUMA_HEADROOM 512 /* YUVRGB_TABLE_LUMA_HEADROOM, swscale_internal.h:39 */
#define PLANE (1024 + 2 * LUMA_HEADROOM)

typedef struct {
  /* C converter: one plane of y_table32 before the per-channel shift, and for every chroma
   * value the ELEMENT offset its table_rV / table_gU / table_bU pointer has from the start of
   * that plane (table_gV is an offset already) */
  uint8_t yval[PLANE];
  int r_idx[256], gu_idx[256], gv_off[256], b_idx[256];
  /* x86 converter: 16-bit lanes of c->yCoeff ... c->yOffset */
  int16_t y_coeff, vr_coeff, ub_coeff, vg_coeff, ug_coeff, y_offset;
} YuvTables;

static int clip_uint8(int64_t v) { return v < 0 ? 0 : v > 255 ? 255 : (int)v; }

/* yuv2rgb.c:762-772 */
static uint16_t round_to_int16(int64_t f) {
  int r = (int)((f + (1 << 15)) >> 16);
  if (r < -0x7FFF) return 0x8000;
  if (r > 0x7FFF) return 0x7FFF;
  return (uint16_t)r;
}

/* yuv2rgb.c:737-749 with elemsize folded out: element offsets instead of byte pointers */
static void fill_table(int *idx, int64_t inc, int y_tab) {
  const int base = y_tab - (int)(inc >> 9);
  for (int i = 0; i < 256; ++i) { /* the head-room entries repeat 0 and 255 (av_clip_uint8) */
    const int64_t cb = (int64_t)i * inc;
    idx[i] = base + (int)(cb >> 16);
  }
}

/* yuv2rgb.c:751-760 */
static void fill_gv_table(int *off_tab, int64_t inc) {
  const int off = -(int)(inc >> 9);
  for (int i = 0; i < 256; ++i) {
    const int64_t cb = (int64_t)i * inc;
    off_tab[i] = off + (int)(cb >> 16);
  }
}

/* ff_yuv2rgb_c_init_tables, yuv2rgb.c:774-855 and the 32-bit case :968-993, for
 * inv_table = ff_yuv2rgb_coeffs[SWS_CS_DEFAULT] (yuv2rgb.c:49-61, row 5), fullRange 0, brightness 0,
 * contrast = saturation = 1 << 16 */
static void init_tables(YuvTables *t) {
  const int inv_table[4] = {104597, 132201, 25675, 53279};
  const int full_range = 0, brightness = 0;
  const int64_t contrast = 1 << 16, saturation = 1 << 16;
  const int yoffs = (full_range ? 384 : 326) + LUMA_HEADROOM;

  int64_t crv = inv_table[0];
  int64_t cbu = inv_table[1];
  int64_t cgu = -inv_table[2];
  int64_t cgv = -inv_table[3];
  int64_t cy = 1 << 16;
  int64_t oy = 0;
  if (!full_range) {
    cy = (cy * 255) / 219;
    oy = 16 << 16;
  } else {
    crv = (crv * 224) / 255;
    cbu = (cbu * 224) / 255;
    cgu = (cgu * 224) / 255;
    cgv = (cgv * 224) / 255;
  }
  cy = (cy * contrast) >> 16;
  crv = (crv * contrast * saturation) >> 32;
  cbu = (cbu * contrast * saturation) >> 32;
  cgu = (cgu * contrast * saturation) >> 32;
  cgv = (cgv * contrast * saturation) >> 32;
  oy -= 256 * brightness;

  t->y_coeff = (int16_t)round_to_int16(cy * (1 << 13));
  t->vr_coeff = (int16_t)round_to_int16(crv * (1 << 13));
  t->ub_coeff = (int16_t)round_to_int16(cbu * (1 << 13));
  t->vg_coeff = (int16_t)round_to_int16(cgv * (1 << 13));
  t->ug_coeff = (int16_t)round_to_int16(cgu * (1 << 13));
  t->y_offset = (int16_t)round_to_int16(oy * (1 << 3));

  /* scale coefficients by cy */
  const int64_t cyd = cy > 1 ? cy : 1;
  crv = ((crv * (1 << 16)) + 0x8000) / cyd;
  cbu = ((cbu * (1 << 16)) + 0x8000) / cyd;
  cgu = ((cgu * (1 << 16)) + 0x8000) / cyd;
  cgv = ((cgv * (1 << 16)) + 0x8000) / cyd;

  int64_t yb = -(384 << 16) - LUMA_HEADROOM * cy - oy;
  for (int i = 0; i < PLANE; ++i) {
    t->yval[i] = (uint8_t)clip_uint8((yb + 0x8000) >> 16);
    yb += cy;
  }
  fill_table(t->r_idx, crv, yoffs);
  fill_table(t->gu_idx, cgu, yoffs);
  fill_table(t->b_idx, cbu, yoffs);
  fill_gv_table(t->gv_off, cgv);
}

static const YuvTables *tables(void) {
  static YuvTables t;
  static int ready;
  if (!ready) {
    init_tables(&t);
    ready = 1;
  }
  return &t;
}

static uint8_t table_at(const YuvTables *t, int idx) {
  if (idx < 0 || idx >= PLANE) abort(); /* the reference would read outside its table */
  return t->yval[idx];
}

/* LOADCHROMA + PUTRGB, yuv2rgb.c:70-81: dst = r[Y] + g[Y] + b[Y] */
static void px_sws_c(const YuvTables *t, int Y, int U, int V, uint8_t *rgb) {
  rgb[0] = table_at(t, t->r_idx[V] + Y);
  rgb[1] = table_at(t, t->gu_idx[U] + t->gv_off[V] + Y);
  rgb[2] = table_at(t, t->b_idx[U] + Y);
}

/* MMX lane arithmetic */
static int16_t wrap16(int v) { return (int16_t)(uint16_t)v; }                  /* psubw, psllw */
static int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static int16_t mulhi(int16_t a, int16_t b) { return (int16_t)(((int32_t)a * b) >> 16); } /* pmulhw */
static uint8_t satu8(int16_t v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }   /* packuswb */

/* x86/yuv2rgb_template.c:84-122 */
static void px_sws_x86(const YuvTables *t, int Y, int U, int V, uint8_t *rgb) {
  const int16_t u = sat16(wrap16(U << 3) - 0x0400); /* psubsw U_OFFSET, yuv2rgb.c:830 */
  const int16_t v = sat16(wrap16(V << 3) - 0x0400);
  const int16_t y = wrap16(wrap16(Y << 3) - t->y_offset); /* psubw */
  const int16_t ug = mulhi(u, t->ug_coeff);
  const int16_t vg = mulhi(v, t->vg_coeff);
  const int16_t yy = mulhi(y, t->y_coeff);
  const int16_t ub = mulhi(u, t->ub_coeff);
  const int16_t vr = mulhi(v, t->vr_coeff);
  const int16_t cg = sat16(ug + vg); /* paddsw */
  rgb[0] = satu8(sat16(vr + yy));
  rgb[1] = satu8(sat16(cg + yy));
  rgb[2] = satu8(sat16(ub + yy));
}

void f360o_yuv_to_rgb_pixel(int model, int Y, int U, int V, uint8_t *rgb) {
  if (model == F360O_YUV_SWS_X86)
    px_sws_x86(tables(), Y, U, V, rgb);
  else
    px_sws_c(tables(), Y, U, V, rgb);
}

void f360o_yuv420p_to_rgb0(uint8_t *dst, int dst_linesize, const uint8_t *y_plane,
                           int y_linesize, const uint8_t *u_plane, int u_linesize,
                           const uint8_t *v_plane, int v_linesize, int width, int height,
                           int model) {
  const YuvTables *t = tables();
  for (int y = 0; y < height; ++y) {
    const uint8_t *py = y_plane + (size_t)y * y_linesize;
    const uint8_t *pu = u_plane + (size_t)(y >> 1) * u_linesize;
    const uint8_t *pv = v_plane + (size_t)(y >> 1) * v_linesize;
    uint8_t *out = dst + (size_t)y * dst_linesize;
    for (int x = 0; x < width; ++x) {
      if (model == F360O_YUV_SWS_X86)
        px_sws_x86(t, py[x], pu[x >> 1], pv[x >> 1], out + 4 * x);
      else
        px_sws_c(t, py[x], pu[x >> 1], pv[x >> 1], out + 4 * x);
      out[4 * x + 3] = 255; /* 255u << abase, yuv2rgb.c:983-984 ; SET_EMPTY_ALPHA, x86/yuv2rgb_template.c:356 */
    }
  }
}
